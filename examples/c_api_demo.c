/* c_api_demo.c -- the C ABI of include/picstep.h used from plain C, with no Python and no torch in the process:
 * what a cgo / JNI / Fortran binding would do.
 *
 *   gcc -std=c99 -O2 -Iinclude examples/c_api_demo.c -o c_api_demo \
 *       -L<csrc> -lpicstep -Wl,-rpath,<csrc> -Wl,-rpath,/opt/rocm/lib
 *   ./c_api_demo particles.bin result.bin [steps]
 *
 * particles.bin: int64 N, int32 Ng, int32 num_envs, double L, double dt, then x[num_envs][N], v[num_envs][N] (float64).
 * result.bin:    x, v [num_envs][N]; n, E_mesh, phi [num_envs][Ng]; KE, PE, PE_reward [num_envs]   (float64).
 * tests/test_gpu_functions.py::test_c_program_drives_the_abi compares the result bit for bit with the Python binding.
 */
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>

#include "picstep.h"

#define CHECK(h, call)                                                                  \
  do {                                                                                  \
    int rc_ = (call);                                                                   \
    if (rc_ != PIC_OK) {                                                                \
      fprintf(stderr, "%s failed (%d): %s\n", #call, rc_, pic_last_error(h));           \
      return 1;                                                                         \
    }                                                                                   \
  } while (0)

int main(int argc, char** argv) {
  if (argc < 3) {
    fprintf(stderr, "usage: %s particles.bin result.bin [steps]\n", argv[0]);
    return 2;
  }
  const int steps = argc > 3 ? atoi(argv[3]) : 10;
  FILE* f = fopen(argv[1], "rb");
  if (!f) { perror(argv[1]); return 2; }
  int64_t N;
  int32_t Ng, E;
  double L, dt;
  if (fread(&N, 8, 1, f) != 1 || fread(&Ng, 4, 1, f) != 1 || fread(&E, 4, 1, f) != 1 || fread(&L, 8, 1, f) != 1 ||
      fread(&dt, 8, 1, f) != 1) { fprintf(stderr, "short header\n"); return 2; }
  const size_t np = (size_t)E * (size_t)N, ng = (size_t)E * (size_t)Ng;
  double* x = malloc(np * 8);
  double* v = malloc(np * 8);
  double* mesh = malloc(3 * ng * 8);
  double* en = malloc(3 * (size_t)E * 8);
  if (!x || !v || !mesh || !en || fread(x, 8, np, f) != np || fread(v, 8, np, f) != np) { fprintf(stderr, "short file\n"); return 2; }
  fclose(f);

  pic_config cfg = {0};                       /* PIC.__init__ arguments that matter to the step (src/env/pic.py:13-27) */
  cfg.N = N; cfg.Ng = Ng; cfg.num_envs = E; cfg.L = L; cfg.n0 = 1.0; cfg.dt = dt; cfg.gamma = 5.0;
  cfg.particle_dtype = PIC_F64; cfg.accum_dtype = PIC_ACC_AUTO; cfg.interpol = PIC_CIC;
  pic_handle* h = NULL;
  CHECK(NULL, pic_create(&cfg, &h));
  CHECK(h, pic_reset(h, x, v, PIC_HOST));              /* PIC.initialize tail (pic.py:76-77) */
  CHECK(h, pic_step(h, NULL, PIC_HOST, steps));        /* steps x PIC.update_state(None) (pic.py:131-146) */
  CHECK(h, pic_get_particles(h, x, v, PIC_HOST));
  CHECK(h, pic_get_fields(h, mesh, mesh + ng, mesh + 2 * ng));
  CHECK(h, pic_get_energies(h, en, en + E, en + 2 * E));
  int64_t bad = -1;
  CHECK(h, pic_bad_count(h, &bad));
  printf("schedule=%s  steps=%d  H[0]=%.17g  PE_reward[0]=%.17g  bad=%lld\n", pic_schedule(h) == 1 ? "resident" : "streaming",
         steps, en[0] + en[E], en[2 * (size_t)E], (long long)bad);
  CHECK(h, pic_destroy(h));

  f = fopen(argv[2], "wb");
  if (!f) { perror(argv[2]); return 2; }
  fwrite(x, 8, np, f); fwrite(v, 8, np, f); fwrite(mesh, 8, 3 * ng, f); fwrite(en, 8, 3 * (size_t)E, f);
  fclose(f);
  free(x); free(v); free(mesh); free(en);
  return bad == 0 ? 0 : 3;
}
