"""Seeded problem generators shared by the parity tests."""
import numpy as np


def random_csr(n, p, mean_nnz, seed, empty_rows=True, values="normal", max_nnz=None):
    """Ragged CSR: row lengths ~ Poisson(mean_nnz) clipped to [0, p] (some rows empty), columns sorted and
    distinct inside a row, float32 values."""
    rng = np.random.default_rng(seed)
    lens = rng.poisson(mean_nnz, n).clip(0 if empty_rows else 1, p if max_nnz is None else min(p, max_nnz))
    if empty_rows and n > 3:
        lens[rng.integers(0, n, max(1, n // 50))] = 0
        lens[rng.integers(0, n, max(1, n // 50))] = 1  # single-nnz rows: pairwise gradient is identically 0
    row_ptr = np.zeros(n + 1, np.int64)
    row_ptr[1:] = np.cumsum(lens)
    col = np.zeros(int(row_ptr[-1]), np.uint32)
    for i in range(n):
        col[row_ptr[i]:row_ptr[i + 1]] = np.sort(rng.choice(p, int(lens[i]), replace=False))
    if values == "ones":
        val = np.ones(len(col), np.float32)
    else:
        val = rng.normal(0, 1, len(col)).astype(np.float32)
    return row_ptr, col, val


def labels(n, seed, task="classification"):
    rng = np.random.default_rng(seed + 7)
    if task == "classification":
        return np.where(rng.random(n) < 0.5, -1.0, 1.0).astype(np.float32)
    return rng.normal(0, 2, n).astype(np.float32)


def params(p, k, seed, stdev=0.1, fp32=True):
    """(w0, w[p], v[k][p]); fp32=True makes every value exactly representable in float32."""
    rng = np.random.default_rng(seed + 13)
    w0 = float(rng.normal(0, 0.1))
    w = rng.normal(0, 0.1, p)
    v = rng.normal(0, stdev, (k, p))
    if fp32:
        w0 = float(np.float32(w0)); w = w.astype(np.float32).astype(np.float64); v = v.astype(np.float32).astype(np.float64)
    return w0, w, v


def rel_err(a, b):
    """normwise relative error max|a-b| / max|b|."""
    a = np.asarray(a, np.float64); b = np.asarray(b, np.float64)
    d = np.max(np.abs(a - b)) if a.size else 0.0
    s = np.max(np.abs(b)) if b.size else 0.0
    return d / s if s > 0 else d


class DevBuf:
    """A device buffer of float64 through the HIP runtime libfmx.so is linked against (ctypes; no torch: importing torch AFTER
    libfmx.so brings a second HIP runtime into the process, see fmwr_amd/distributed.py)."""
    _hip = None

    @classmethod
    def hip(cls):
        if cls._hip is None:
            import ctypes as C
            from fmwr_amd import _lib
            _lib.lib()  # libfmx.so first: its libamdhip64.so.7 is then the one dlopen hands back
            cls._hip = C.CDLL("libamdhip64.so.7")
        return cls._hip

    def __init__(self, n, dtype=np.float64):
        import ctypes as C
        self.n, self.dtype = int(n), np.dtype(dtype)
        self.ptr = C.c_void_p()
        rc = self.hip().hipMalloc(C.byref(self.ptr), C.c_size_t(max(1, self.n) * self.dtype.itemsize))
        if rc != 0:
            raise MemoryError(f"hipMalloc failed ({rc})")

    @classmethod
    def from_numpy(cls, a):
        a = np.ascontiguousarray(a)
        b = cls(a.size, a.dtype)
        b.upload(a)
        return b

    def upload(self, a, offset=0):
        import ctypes as C
        a = np.ascontiguousarray(a, self.dtype)
        rc = self.hip().hipMemcpy(C.c_void_p(self.ptr.value + offset * self.dtype.itemsize), a.ctypes.data_as(C.c_void_p), C.c_size_t(a.nbytes), C.c_int(1))
        assert rc == 0, rc

    def numpy(self):
        import ctypes as C
        out = np.empty(self.n, self.dtype)
        rc = self.hip().hipMemcpy(out.ctypes.data_as(C.c_void_p), self.ptr, C.c_size_t(out.nbytes), C.c_int(2))
        assert rc == 0, rc
        return out

    def free(self):
        if self.ptr:
            self.hip().hipFree(self.ptr)
            self.ptr = None

    def __del__(self):
        try:
            self.free()
        except Exception:
            pass
