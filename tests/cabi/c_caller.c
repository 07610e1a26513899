/* TEST INFRASTRUCTURE: a plain-C caller of include/battgp.h - the drop-in boundary used the way a C host (or any FFI that
 * speaks the C ABI) would use it: create -> set_kernel -> fit -> predict -> refit -> lml_grad -> destroy, on the known
 * answers of the reference's own unit test (tests/gp/test_standard_models.py:14-47: one training point y = 10 at x = 1 with
 * noise = outputscale = 3 gives mean 5, variance 1.5; the same point twice gives mean 20/3, variance 1) and on a small
 * production-kernel problem whose LML is handed in by the test.
 *
 *   gcc -std=c99 -pedantic -Wall -Wextra -Werror -Iinclude tests/cabi/c_caller.c -L<dir> -l<lib> -lm
 *
 * Exit code 0 = every check passed; 77 = the library reported "no device" at bgp_create (the product library on a box
 * without a GPU: it must say so instead of computing anything); anything else = failure (message on stderr). */
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "battgp.h"

#define CHECK(cond, ...)                          \
  do {                                            \
    if (!(cond)) {                                \
      fprintf(stderr, "c_caller: " __VA_ARGS__);  \
      fprintf(stderr, "\n");                      \
      return 1;                                   \
    }                                             \
  } while (0)

static int near(double a, double b, double rel) { return fabs(a - b) <= rel * fmax(fabs(b), 1e-300); }

int main(int argc, char** argv) {
  bgp_handle* h = NULL;
  int rc = bgp_create(&h, 0);
  if (rc != 0) {
    fprintf(stderr, "c_caller: bgp_create failed (%d): %s\n", rc, bgp_last_error(NULL)); /* NULL: the message of the failed create */
    return 77;
  }
  CHECK(bgp_version() >= 100, "version %d", bgp_version());

  /* --- known answers of tests/gp/test_standard_models.py --- */
  {
    const double hyp[3] = {3.0, 3.0, 2.0}; /* noise, outputscale, lengthscale */
    const double x1[1] = {1.0}, y1[1] = {10.0}, xq[1] = {1.0};
    const double x2[2] = {1.0, 1.0}, y2[2] = {10.0, 10.0};
    double lml = 0.0, jit = -1.0, mean = 0.0, var = 0.0;
    CHECK(bgp_set_kernel(h, BGP_KERNEL_SCALED_RBF, hyp, 3) == 0, "set_kernel: %s", bgp_last_error(h));
    CHECK(bgp_fit(h, x1, y1, 1, 1, &lml, &jit) == 0, "fit: %s", bgp_last_error(h));
    CHECK(jit == 0.0, "jitter %g", jit);
    /* log N(10; 0, 6) */
    CHECK(near(lml, -0.5 * 100.0 / 6.0 - 0.5 * log(6.0) - 0.5 * log(2.0 * acos(-1.0)), 1e-12), "lml %.15g", lml);
    CHECK(bgp_predict(h, xq, 1, &mean, &var, -1.0) == 0, "predict: %s", bgp_last_error(h));
    CHECK(near(mean, 5.0, 1e-12) && near(var, 1.5, 1e-12), "one point: mean %.15g var %.15g", mean, var);
    CHECK(bgp_fit(h, x2, y2, 2, 1, &lml, &jit) == 0, "fit 2: %s", bgp_last_error(h));
    CHECK(bgp_predict(h, xq, 1, &mean, &var, -1.0) == 0, "predict 2: %s", bgp_last_error(h));
    CHECK(near(mean, 20.0 / 3.0, 1e-12) && near(var, 1.0, 1e-12), "two points: mean %.15g var %.15g", mean, var);
  }

  /* --- error behaviour: wrong hyper-parameter count is an argument error (< 0) with a message, never a crash --- */
  {
    const double bad[2] = {1.0, 1.0};
    const double x[4] = {0.0, 1.0, 2.0, 3.0}, y[1] = {1.0};
    double lml, jit;
    int r = bgp_set_kernel(h, BGP_KERNEL_BATTGP, bad, 2);
    if (r == 0) r = bgp_fit(h, x, y, 1, 4, &lml, &jit);
    CHECK(r < 0 && strlen(bgp_last_error(h)) > 0, "bad hyper-parameter vector accepted (rc %d)", r);
  }

  /* --- production kernel, data and expected LML from the command line: N D lml x[N*D] y[N] hyp[3+D-1] --- */
  if (argc > 3) {
    const int n = atoi(argv[1]), d = atoi(argv[2]);
    const double want = atof(argv[3]);
    const int nhyp = 3 + d - 1;
    CHECK(argc == 4 + n * d + n + nhyp, "expected %d arguments, got %d", 4 + n * d + n + nhyp, argc);
    double* x = (double*)malloc(sizeof(double) * (size_t)(n * d));
    double* y = (double*)malloc(sizeof(double) * (size_t)n);
    double hyp[BGP_MAX_HYP], grad[BGP_MAX_HYP], lml = 0.0, lml2 = 0.0, jit = 0.0;
    int i, a = 4;
    for (i = 0; i < n * d; ++i) x[i] = atof(argv[a++]);
    for (i = 0; i < n; ++i) y[i] = atof(argv[a++]);
    for (i = 0; i < nhyp; ++i) hyp[i] = atof(argv[a++]);
    CHECK(bgp_set_kernel(h, BGP_KERNEL_BATTGP, hyp, nhyp) == 0, "set_kernel K0: %s", bgp_last_error(h));
    CHECK(bgp_fit(h, x, y, n, d, &lml, &jit) == 0, "fit K0: %s", bgp_last_error(h));
    CHECK(near(lml, want, 1e-6), "K0 lml %.15g, expected %.15g", lml, want);
    /* the optimiser's call pair: re-fit on the resident data, then the gradient (src/gp/training.py:39-41) */
    CHECK(bgp_refit(h, hyp, nhyp, &lml2, &jit) == 0, "refit: %s", bgp_last_error(h));
    CHECK(lml2 == lml, "refit at the same point: %.17g vs %.17g", lml2, lml);
    CHECK(bgp_lml_grad(h, grad, nhyp) == 0, "lml_grad: %s", bgp_last_error(h));
    for (i = 0; i < nhyp; ++i) CHECK(grad[i] == grad[i], "gradient component %d is NaN", i);
    printf("lml %.17g grad", lml);
    for (i = 0; i < nhyp; ++i) printf(" %.17g", grad[i]);
    printf("\n");
    free(x);
    free(y);
  }
  bgp_destroy(h);
  printf("c_caller ok\n");
  return 0;
}
