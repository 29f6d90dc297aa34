"""Known-answer data of SURVEY.md Appendix B: the reference's own learners (headers compiled
unmodified in the survey session) on a 6x5 CSR matrix.  Data, not code: inputs + expected outputs."""
import math

import numpy as np

ROW_PTR = np.array([0, 2, 4, 7, 8, 10, 12], np.int64)
COL = np.array([0, 3, 1, 2, 0, 2, 4, 3, 1, 4, 0, 1], np.uint32)
VAL = np.array([1, .5, 2, 1, 1, -1, .25, 3, 1, 1, .5, .5], np.float32)
Y = np.array([1, -1, 1, -1, 1, -1], np.float32)
N, P_FEAT, K = 6, 5, 3
INIT_STDEV = 0.1
L2_REGW, L2_REGV = 0.01, 0.02
MAX_ITER, TRACE_STEP = 50, 10

SGD_W0 = -0.10186945032924256
SGD_W = [0.10246702307860416, -0.23372389731408191, -0.40782913304270302, -0.46911651890344552, 0.31280666386882444]
SGD_V0 = [0.010527388918283573, 0.086610025327578929, -0.15601382906491748, -0.052232816248671023, 0.046943557687816662]
SGD_LL = [-4.087788511, -3.760688163, -3.531096602, -3.356440428, -3.215448292, -3.105536052]
FTRL_W0 = -0.10237722059750001
FTRL_LL = [-4.097225536, -3.789926011, -3.597139176, -3.45248008, -3.336440102, -3.249096699]
TDAP_W0 = -0.22442994088195486   # TDAP_Learner class defaults: gamma = 0.001, alpha_w = alpha_v = 0.1, no L1
TDAP_LL = [-3.937315206, -3.68040471, -3.296405521, -2.9887651, -2.773177912, -2.613076344]
TDAP_GAMMA = 0.001
TRACE_ITERS = [0, 10, 20, 30, 40, 49]


def harness_v0():
    """V0 as the survey harness drew it: Rf_rnorm(mu, sd) = mu + sd*sqrt(-2 ln a)*cos(2 pi b), a,b successive
    draws of the LCG s = s*6364136223846793005 + 1442695040888963407 mod 2^64, u = ((s>>11)+0.5)/2^53,
    s0 = 12345, filled in [f][j] memory order (util/Dmatrix.h:143-146)."""
    s = 12345
    mask = (1 << 64) - 1

    def u():
        nonlocal s
        s = (s * 6364136223846793005 + 1442695040888963407) & mask
        return ((s >> 11) + 0.5) / float(1 << 53)

    v = np.zeros(K * P_FEAT)
    for i in range(K * P_FEAT):
        a = u(); b = u()
        v[i] = 0.0 + INIT_STDEV * math.sqrt(-2.0 * math.log(a)) * math.cos(2.0 * math.pi * b)
    return v
