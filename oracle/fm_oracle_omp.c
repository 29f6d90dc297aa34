/* All-core CPU baselines for bench.py's report ("CPU baseline beside it", item ii).  TEST
 * INFRASTRUCTURE like the rest of oracle/: timed beside the GPU numbers, never part of the product.
 *
 * The reference has a parallel FORWARD (Model::predict_batch, OpenMP over rows, core/Model.h:106-161) and no parallel
 * training at all (todo_list.md:7).  So:
 *   fmo_omp_predict_batch -- the reference's forward with its own parallelisation (rows across threads);
 *   fmo_omp_sgd_hogwild   -- what a CPU user would do to use all cores: the reference's example step
 *                            (SGD_Learner.h:92-138) run lock-free over row ranges (Hogwild).  Not deterministic and not
 *                            a parity object: a throughput yardstick only.
 * Built separately (-fopenmp) into _build/libfm_oracle_omp.so; the serial oracle stays free of OpenMP. */
#include <omp.h>

#include "fm_oracle.c"

void fmo_omp_predict_batch(const fmo_params* P, uint32_t p, double w0, const double* w, const double* v, const fmo_csr* X,
                           double* out, int threads) {
  const int k = P->k;
#pragma omp parallel num_threads(threads)
  {
    double* m_sum = (double*)calloc((size_t)(k ? k : 1) * 2, sizeof(double));
    double* m_sum_sqr = m_sum + (k ? k : 1);
#pragma omp for schedule(static)
    for (int64_t i = 0; i < X->n; ++i) out[i] = fmo_predict(P, p, w0, w, v, X, i, m_sum, m_sum_sqr);
    free(m_sum);
  }
}

/* one lock-free pass over rows [0, n); returns the examples processed */
int64_t fmo_omp_sgd_hogwild(const fmo_params* P, uint32_t p, double* w0, double* w, double* v, const fmo_csr* X,
                            const float* y, int threads) {
  int l1_penalty; double regw, regv;
  fmo_sgd_mode(P, &l1_penalty, &regw, &regv);
  if (l1_penalty) return -1; /* the cumulative-penalty state is inherently serial */
  const int k = P->k;
#pragma omp parallel num_threads(threads)
  {
    double* m_sum = (double*)calloc((size_t)(k ? k : 1) * 2, sizeof(double));
    double* m_sum_sqr = m_sum + (k ? k : 1);
    double u_w = 0.0, u_v = 0.0;
#pragma omp for schedule(static)
    for (int64_t i = 0; i < X->n; ++i)
      fmo_sgd_example(P, p, w0, w, v, X, y, i, 0, regw, regv, NULL, NULL, &u_w, &u_v, m_sum, m_sum_sqr);
    free(m_sum);
  }
  return X->n;
}

int fmo_omp_max_threads(void) { return omp_get_max_threads(); }
