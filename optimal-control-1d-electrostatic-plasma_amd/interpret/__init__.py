from .spectrum import compute_E_k_spectrum
