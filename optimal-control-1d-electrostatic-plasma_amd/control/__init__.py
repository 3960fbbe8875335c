from .actuator import E_field
from .reward import Reward
