/* A plain C caller of include/fmx.h: the call sequence FM() / FMPredict() make around the learner seam (src/FM.cpp:31-161,
 * :177-214), on the 6 x 5 matrix of SURVEY.md Appendix B.  It trains the reference's SGD learner through the C ABI and
 * prints w0, w and the training log-likelihood; tests/test_gpu_c_caller.py compares them with the reference's own known
 * answers (Appendix B).  No Python, no C++, no torch on this side of the boundary.
 *   gcc -std=c99 -Iinclude examples/kat_c.c -Lfmwr_amd -lfmx -Wl,-rpath,$PWD/fmwr_amd -lm -o kat_c && ./kat_c */
#include <math.h>
#include <stdint.h>
#include <stdio.h>

#include "fmx.h"

#define CHECK(call) do { if ((call) != FMX_OK) { fprintf(stderr, "%s: %s\n", #call, fmx_last_error()); return 1; } } while (0)

int main(void) {
  /* the R list fm.matrix() builds (R/fm_matrix.R:25-34): value, 0-based col_idx, row_size, dim */
  const double value[12] = {1, .5, 2, 1, 1, -1, .25, 3, 1, 1, .5, .5};
  const int32_t col_idx[12] = {0, 3, 1, 2, 0, 2, 4, 3, 1, 4, 0, 1};
  const int32_t row_size[6] = {2, 2, 3, 1, 2, 2};
  const double labels[6] = {1, -1, 1, -1, 1, -1};
  enum { N = 6, P = 5, K = 3 };

  /* V0 as the survey harness drew it (its Rf_rnorm: Box-Muller on a 64-bit LCG, s0 = 12345), [f][j] order */
  double v_fj[K * P], v_kxp[K * P];
  uint64_t s = 12345;
  for (int i = 0; i < K * P; ++i) {
    s = s * 6364136223846793005ull + 1442695040888963407ull; const double a = ((double)(s >> 11) + 0.5) / 9007199254740992.0;
    s = s * 6364136223846793005ull + 1442695040888963407ull; const double b = ((double)(s >> 11) + 0.5) / 9007199254740992.0;
    v_fj[i] = 0.1 * sqrt(-2.0 * log(a)) * cos(2.0 * 3.14159265358979323846 * b);
  }
  for (int f = 0; f < K; ++f) for (int j = 0; j < P; ++j) v_kxp[f + j * K] = v_fj[f * P + j]; /* the R k x p matrix, column-major */

  fmx_config c;
  CHECK(fmx_config_default(&c));
  c.task = FMX_TASK_CLASSIFICATION; c.solver = FMX_SOLVER_SGD; c.num_factor = K;
  c.l2_w1 = 0.01; c.l2_v = 0.02; c.learn_rate = 0.05; c.random_step = 1;
  c.mode = FMX_MODE_SEQUENTIAL; c.min_target = -1.0; c.max_target = 1.0;

  fmx_engine* e = NULL; fmx_matrix* m = NULL;
  CHECK(fmx_engine_create(&c, P, &e));
  CHECK(fmx_matrix_from_rlist(0, N, P, 12, value, col_idx, row_size, labels, &m));
  double w0 = 0.0, w[P] = {0}, v[K * P];
  CHECK(fmx_set_params(e, 0.0, w, v_kxp));
  int64_t done = 0;
  CHECK(fmx_train(e, m, 50, &done));                       /* learner->learn(data), max_iter = 50 */
  CHECK(fmx_get_params(e, &w0, w, v));
  double ll = 0.0;
  CHECK(fmx_evaluate(e, m, FMX_EVAL_LL, &ll));             /* Tracker::report on the training data */
  double prob[N];
  CHECK(fmx_predict(e, m, prob, FMX_LINK_LOGISTIC));       /* FMPredict */
  printf("examples %lld\nw0 %.17g\n", (long long)done, w0);
  for (int j = 0; j < P; ++j) printf("lin%d %.17g\n", j, w[j]);
  for (int j = 0; j < P; ++j) printf("v0_%d %.17g\n", j, v[0 + j * K]);
  printf("ll %.12g\n", ll);
  for (int i = 0; i < N; ++i) printf("p%d %.17g\n", i, prob[i]);
  CHECK(fmx_matrix_destroy(m));
  CHECK(fmx_engine_destroy(e));

  /* The multi-GPU entry from C: cfg.n_gpus = 2 (what options("FM.threads") = 2 becomes, src/FM.cpp:59,97) with both replicas on
   * one device (gpus_share_device: a one-GPU box).  fmx_train cuts the 6 rows into two shards of 3; a global step is 2 rows of
   * each shard.  The same steps on ONE engine are the rows in the order 0 1 3 4 | 2 5 with 4 rows per step. */
  {
    fmx_config g;
    CHECK(fmx_config_default(&g));
    g.task = FMX_TASK_CLASSIFICATION; g.solver = FMX_SOLVER_SGD; g.num_factor = K;
    g.l2_w1 = 0.01; g.l2_v = 0.02; g.learn_rate = 0.05;
    g.mode = FMX_MODE_MINIBATCH; g.state_fp64 = 1; g.min_target = -1.0; g.max_target = 1.0;
    const int perm[N] = {0, 1, 3, 4, 2, 5};
    double value2[12], labels2[N]; int32_t col2[12], size2[N];
    int start[N], at = 0;
    for (int i = 0, t = 0; i < N; ++i) { start[i] = t; t += row_size[i]; }
    for (int i = 0; i < N; ++i) {
      const int r = perm[i];
      size2[i] = row_size[r]; labels2[i] = labels[r];
      for (int q = 0; q < row_size[r]; ++q, ++at) { value2[at] = value[start[r] + q]; col2[at] = col_idx[start[r] + q]; }
    }
    for (int form = 0; form < 2; ++form) {
      fmx_config cc = g;
      cc.n_gpus = form == 0 ? 2 : 1; cc.gpus_share_device = form == 0 ? 1 : 0;
      cc.batch_rows = form == 0 ? 2 : 4;                     /* rows per step PER GPU */
      fmx_engine* ge = NULL; fmx_matrix* gm = NULL;
      CHECK(fmx_engine_create(&cc, P, &ge));
      if (form == 0) CHECK(fmx_matrix_from_rlist(0, N, P, 12, value, col_idx, row_size, labels, &gm));
      else CHECK(fmx_matrix_from_rlist(0, N, P, 12, value2, col2, size2, labels2, &gm));
      double zero[P] = {0};
      CHECK(fmx_set_params(ge, 0.0, zero, v_kxp));
      int64_t gdone = 0;
      CHECK(fmx_train(ge, gm, 18, &gdone));                  /* three passes */
      double gw0 = 0.0, gw[P], gv[K * P];
      CHECK(fmx_get_params(ge, &gw0, gw, gv));
      printf("g%d_examples %lld\ng%d_w0 %.17g\n", cc.n_gpus, (long long)gdone, cc.n_gpus, gw0);
      for (int j = 0; j < P; ++j) printf("g%d_lin%d %.17g\n", cc.n_gpus, j, gw[j]);
      for (int j = 0; j < K * P; ++j) printf("g%d_v%d %.17g\n", cc.n_gpus, j, gv[j]);
      if (form == 0 && fmx_step(ge, gm, 0, 0) == FMX_OK) { fprintf(stderr, "fmx_step on an n_gpus = 2 handle must be refused\n"); return 1; }
      CHECK(fmx_matrix_destroy(gm));
      CHECK(fmx_engine_destroy(ge));
    }
  }
  return 0;
}
