"""Thin object wrappers over the C ABI handles (fmx_engine, fmx_matrix).  numpy in, numpy out."""
import ctypes as C

import numpy as np

from . import _lib as L


def _p(a):
    return a.ctypes.data_as(C.c_void_p) if a is not None else None


def fields_spec(n_dense, field_vocab, skew, seed):
    """(fmx_fields_spec, the numpy array it points to -- keep it alive for the call)."""
    vocab = np.ascontiguousarray(field_vocab, np.uint32)
    spec = L.FieldsSpec(C.sizeof(L.FieldsSpec), int(n_dense), len(vocab), 0, vocab.ctypes.data, float(skew), int(seed))
    return spec, vocab


# BASELINE.json configs[3]: 13 dense + 26 categorical fields, 33 000 000 features in all.  Vocabulary sizes follow the spread
# of the Criteo click logs' 26 categorical columns (a few fields of millions of values, many of a handful).
CRITEO_VOCAB = [9_900_000, 7_900_000, 6_700_000, 4_900_000, 2_200_000, 590_000, 400_000, 250_000, 72404, 39_000, 17_000, 12_000, 7_400, 7_100,
                2_200, 1_500, 980, 160, 110, 63, 36, 14, 10, 4, 3, 3]
assert 13 + sum(CRITEO_VOCAB) == 33_000_000


class Matrix:
    """A device-resident fm.matrix (fmx_matrix*)."""

    def __init__(self, handle, n, p, nnz):
        self.h, self.n, self.p, self.nnz = handle, n, p, nnz

    @classmethod
    def _wrap(cls, handle):
        n, p, nnz = C.c_int64(), C.c_uint32(), C.c_int64()
        L.check(L.lib().fmx_matrix_info(handle, C.byref(n), C.byref(p), C.byref(nnz)))
        return cls(handle, n.value, p.value, nnz.value)

    @classmethod
    def from_csr(cls, row_ptr, col, val, p, y=None, device=0):
        row_ptr = np.ascontiguousarray(row_ptr, np.int64)
        col = np.ascontiguousarray(col, np.uint32)
        val = np.ascontiguousarray(val, np.float32)
        y = None if y is None else np.ascontiguousarray(y, np.float32)
        n = len(row_ptr) - 1
        if len(col) != len(val) or (n >= 0 and len(col) != int(row_ptr[-1])):
            raise ValueError("col/val length does not match row_ptr")
        if y is not None and len(y) != n:
            raise ValueError("target's length is not equal the number of cases...")  # core/Data.h:75-76
        h = C.c_void_p()
        L.check(L.lib().fmx_matrix_from_csr(C.c_int(device), C.c_int64(n), C.c_uint32(p), _p(row_ptr), _p(col), _p(val), _p(y),
                                            C.byref(h)))
        return cls._wrap(h)

    @classmethod
    def from_rlist(cls, value, col_idx, row_size, p, labels=None, device=0):
        """R's fm.matrix$features layout (R/fm_matrix.R:25-34)."""
        value = np.ascontiguousarray(value, np.float64)
        col_idx = np.ascontiguousarray(col_idx, np.int32)
        row_size = np.ascontiguousarray(row_size, np.int32)
        labels = None if labels is None else np.ascontiguousarray(labels, np.float64)
        h = C.c_void_p()
        L.check(L.lib().fmx_matrix_from_rlist(C.c_int(device), C.c_int64(len(row_size)), C.c_uint32(p), C.c_int64(len(value)),
                                              _p(value), _p(col_idx), _p(row_size), _p(labels), C.byref(h)))
        return cls._wrap(h)

    @classmethod
    def from_dgc(cls, x, i, p, nrow, ncol, labels=None, device=0):
        """A dgCMatrix's slots as they lie in R (x, i = 0-based row indices, p = column pointers): transposed to rows on the device (fmx_matrix_from_dgc;
        R/fm_matrix.R:26-33 does it with Matrix::t on the host).  scipy: `c = scipy.sparse.csc_matrix(...); from_dgc(c.data, c.indices, c.indptr, *c.shape)`."""
        x = np.ascontiguousarray(x, np.float64)
        i = np.ascontiguousarray(i, np.int32)
        p = np.ascontiguousarray(p, np.int32)
        if len(p) != ncol + 1:
            raise ValueError("p must hold ncol + 1 column pointers")
        if len(i) != len(x):
            raise ValueError("the lengths of i and x differ")
        labels = None if labels is None else np.ascontiguousarray(labels, np.float64)
        if labels is not None and len(labels) != nrow:
            raise ValueError("target's length is not equal the number of cases...")
        h = C.c_void_p()
        L.check(L.lib().fmx_matrix_from_dgc(C.c_int(device), C.c_int64(nrow), C.c_uint32(ncol), C.c_int64(len(x)), _p(x), _p(i), _p(p), _p(labels), C.byref(h)))
        return cls._wrap(h)

    @classmethod
    def synthetic(cls, n, p, nnz_per_row, seed, row_offset=0, device=0):
        h = C.c_void_p()
        L.check(L.lib().fmx_matrix_synthetic(C.c_int(device), C.c_int64(n), C.c_uint32(p), C.c_int32(nnz_per_row), C.c_uint64(seed),
                                             C.c_int64(row_offset), C.byref(h)))
        return cls._wrap(h)

    @classmethod
    def synthetic_iid(cls, n, p, nnz_per_row, seed, law=L.COLUMNS_UNIFORM, zipf_s=1.05, row_offset=0, device=0):
        """i.i.d. columns (uniform or Zipf), sorted inside the row (fmx_matrix_synthetic_iid)."""
        h = C.c_void_p()
        L.check(L.lib().fmx_matrix_synthetic_iid(C.c_int(device), C.c_int64(n), C.c_uint32(p), C.c_int32(nnz_per_row), C.c_uint64(seed), C.c_int64(row_offset),
                                                 C.c_int32(law), C.c_double(zipf_s), C.byref(h)))
        return cls._wrap(h)

    @classmethod
    def synthetic_ragged(cls, n, p, mean_nnz, seed, min_nnz=1, max_nnz=64, row_offset=0, device=0):
        """Row lengths Poisson(mean_nnz) clipped to [min_nnz, max_nnz], i.i.d. uniform sorted columns (fmx_matrix_synthetic_ragged: SURVEY 8(d)'s ragged variant)."""
        h = C.c_void_p()
        L.check(L.lib().fmx_matrix_synthetic_ragged(C.c_int(device), C.c_int64(n), C.c_uint32(p), C.c_double(mean_nnz), C.c_int32(min_nnz), C.c_int32(max_nnz),
                                                    C.c_uint64(seed), C.c_int64(row_offset), C.byref(h)))
        return cls._wrap(h)

    @classmethod
    def synthetic_fields(cls, n, n_dense, field_vocab, skew, seed, row_offset=0, device=0):
        """Criteo-shaped rows: n_dense always-present features + one feature of each categorical field (fmx_matrix_synthetic_fields)."""
        spec, keep = fields_spec(n_dense, field_vocab, skew, seed)
        h = C.c_void_p()
        L.check(L.lib().fmx_matrix_synthetic_fields(C.c_int(device), C.c_int64(n), C.byref(spec), C.c_int64(row_offset), C.byref(h)))
        return cls._wrap(h)

    def set_fields(self, n_dense, field_base):
        """Vouch for a field layout (fmx_matrix_set_fields): n_dense always-present columns, then one id of every field c in
        [field_base[c], field_base[c + 1]) with value 1; checked on the device."""
        fb = np.ascontiguousarray(field_base, np.uint32)
        L.check(L.lib().fmx_matrix_set_fields(self.h, C.c_int32(int(n_dense)), C.c_int32(len(fb) - 1), _p(fb)))

    def synthetic_values(self, seed, row_offset=0):
        """Redraw every stored value uniform in (0, 1) on the device (fmx_matrix_synthetic_values: SURVEY 8(d)'s value variant); the matrix stops being one-hot."""
        L.check(L.lib().fmx_matrix_synthetic_values(self.h, C.c_uint64(seed), C.c_int64(row_offset)))
        return self

    def set_labels(self, y):
        y = np.ascontiguousarray(y, np.float32)
        if len(y) != self.n:
            raise ValueError("target's length is not equal the number of cases...")
        L.check(L.lib().fmx_matrix_set_labels(self.h, _p(y)))

    def scales(self, norm_columns):
        """SMatrix::scales: z-score the listed columns in place; returns (mean[p], std[p])."""
        nc = np.ascontiguousarray(norm_columns, np.int32)
        mean = np.zeros(max(self.p, 1)); std = np.zeros(max(self.p, 1))
        L.check(L.lib().fmx_matrix_scales(self.h, _p(nc), C.c_int64(len(nc)), _p(mean), _p(std)))
        return mean[: self.p], std[: self.p]

    def normalize(self, mean, std):
        """SMatrix::normalize: apply a model's Scales."""
        mean = np.ascontiguousarray(mean, np.float64); std = np.ascontiguousarray(std, np.float64)
        if len(mean) != self.p or len(std) != self.p:
            raise ValueError("the length of scale:mean or scale:std is not equal")  # util/Smatrix.h:143-145
        L.check(L.lib().fmx_matrix_normalize(self.h, _p(mean), _p(std)))

    def rows_form(self):
        """0 / 1 / 2: the form of phase 1 that large steps take on this matrix (fmx_matrix_rows_form: per-row lane groups, flat, pulled)."""
        f = C.c_int32()
        L.check(L.lib().fmx_matrix_rows_form(self.h, C.byref(f)))
        return f.value

    def export(self, r0=0, r1=None):
        """rows [r0, r1) -> (row_ptr, col, val, y) numpy arrays (row_ptr rebased to 0)."""
        r1 = self.n if r1 is None else r1
        rp = np.zeros(r1 - r0 + 1, np.int64)
        L.check(L.lib().fmx_matrix_export(self.h, C.c_int64(r0), C.c_int64(r1), _p(rp), None, None, None))
        cnt = int(rp[-1])
        col = np.zeros(max(cnt, 1), np.uint32); val = np.zeros(max(cnt, 1), np.float32); y = np.zeros(max(r1 - r0, 1), np.float32)
        L.check(L.lib().fmx_matrix_export(self.h, C.c_int64(r0), C.c_int64(r1), _p(rp), _p(col), _p(val), _p(y)))
        return rp, col[:cnt], val[:cnt], y[: r1 - r0]

    def close(self):
        if self.h:
            L.lib().fmx_matrix_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class Source:
    """fmx_source: generated rows handed out one step at a time (each a Matrix of one batch, owned by the source)."""

    def __init__(self, engine, total_rows, nnz_per_row=0, seed=0, row_offset=0, fields=None):
        spec = None
        if fields is not None:
            spec, self._keep = fields_spec(fields[0], fields[1], fields[2], seed)
        self.h = C.c_void_p()
        L.check(L.lib().fmx_source_open(engine.h, C.byref(spec) if spec is not None else None, C.c_int32(nnz_per_row), C.c_uint64(seed), C.c_int64(row_offset),
                                        C.c_int64(total_rows), C.byref(self.h)))
        self.batch_rows = int(engine.cfg.batch_rows)
        self.steps = -(-int(total_rows) // self.batch_rows)
        # fmx_source holds a raw fmx_engine* and fmx_source_close waits on that engine's stream: the engine must outlive the source.  The source keeps
        # it alive (so `Engine(...).source(...)` and interpreter shutdown order are safe) and registers itself so that Engine.close() closes it first.
        self._engine = engine
        engine._sources.append(self)

    def next(self):
        """the next step's Matrix (a borrowed handle: never close it; it holds the source alive) or None at the end"""
        if not self.h:
            raise ValueError("the source is closed")
        mh, rows = C.c_void_p(), C.c_int64()
        L.check(L.lib().fmx_source_next(self.h, C.byref(mh), C.byref(rows)))
        if not mh.value:
            return None
        m = Matrix._wrap(mh)
        m.close = lambda: None   # owned by the source
        m.__dict__["_borrowed"] = True
        m.__dict__["_source"] = self
        return m

    def close(self):
        """waits for the engine's stream; returns the host seconds spent waiting for tiles' counts"""
        wait = C.c_double()
        if self.h:
            h, self.h = self.h, None
            eng, self._engine = self._engine, None
            if eng is not None and self in eng._sources:
                eng._sources.remove(self)
            L.check(L.lib().fmx_source_close(h, C.byref(wait)))
        return wait.value

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def measure_gather(table_bytes, row_bytes, n_groups=262_144, per_group=32, in_flight=4, reps=30, device=0):
    """rows/s the memory system serves for uniformly random rows of `row_bytes` from a `table_bytes` table (fmx_measure_gather)."""
    out = C.c_double()
    L.check(L.lib().fmx_measure_gather(C.c_int(device), C.c_int64(table_bytes), C.c_int32(row_bytes), C.c_int64(n_groups), C.c_int32(per_group),
                                       C.c_int32(in_flight), C.c_int32(reps), C.byref(out)))
    return out.value


def measure_gather_matrix(m, r0, nrows, table_rows, row_bytes, in_flight=4, reps=20):
    """rows/s of the bare gather of the table rows that rows [r0, r0 + nrows) of matrix `m` name (fmx_measure_gather_matrix)."""
    out = C.c_double()
    L.check(L.lib().fmx_measure_gather_matrix(m.h, C.c_int64(r0), C.c_int64(nrows), C.c_int64(table_rows), C.c_int32(row_bytes), C.c_int32(in_flight),
                                              C.c_int32(reps), C.byref(out)))
    return out.value


class Engine:
    """Parameters + optimizer state on one GPU (fmx_engine*)."""

    def __init__(self, num_features, **kw):
        cfg = L.default_config()
        for key, value in kw.items():
            if not hasattr(cfg, key):
                raise TypeError(f"unknown engine option {key!r}")
            setattr(cfg, key, value)
        self.cfg = cfg
        self.p = int(num_features)
        self.k = int(cfg.num_factor)
        self.h = C.c_void_p()
        self._sources = []   # open Source objects (they point at this engine)
        L.check(L.lib().fmx_engine_create(C.byref(cfg), C.c_uint64(self.p), C.byref(self.h)))

    def k_padded(self):
        """padded factor count of the mini-batch tables (4 * 2^m floats, or 2 * 2^m doubles with state_fp64)"""
        kp = 2 if self.cfg.state_fp64 else 4
        while kp < self.k:
            kp *= 2
        return kp

    def set_params(self, w0=0.0, w=None, v=None):
        """v: (k, p) array as R holds it (k x p matrix); stored column-major for the ABI."""
        w = None if w is None else np.ascontiguousarray(w, np.float64)
        vv = None
        if v is not None:
            v = np.asarray(v, np.float64)
            if v.shape != (self.k, self.p):
                raise ValueError(f"v must have shape ({self.k}, {self.p})")
            vv = np.ascontiguousarray(v.T).ravel()  # element (f, j) at f + j*k
        if w is not None and w.shape != (self.p,):
            raise ValueError(f"w must have shape ({self.p},)")
        L.check(L.lib().fmx_set_params(self.h, C.c_double(w0), _p(w), _p(vv)))

    def get_params(self):
        w0 = C.c_double()
        w = np.zeros(self.p)
        vv = np.zeros(max(self.k * self.p, 1))
        L.check(L.lib().fmx_get_params(self.h, C.byref(w0), _p(w), _p(vv)))
        v = vv[: self.k * self.p].reshape(self.p, self.k).T.copy()
        return w0.value, w, v

    def get_w0(self):
        """the global bias alone (waits for the engine's stream; no table is copied)"""
        w0 = C.c_double()
        L.check(L.lib().fmx_get_params(self.h, C.byref(w0), None, None))
        return w0.value

    def init_normal(self, seed, mean=0.0, stdev=0.01):
        """w0 = 0, w = 0, V ~ N(mean, stdev) drawn on the device (synthetic workloads; not R's generator)."""
        L.check(L.lib().fmx_init_normal(self.h, C.c_uint64(seed), C.c_double(mean), C.c_double(stdev)))

    def get_rows(self, ids):
        """(w[n], v[k][n]) of the listed features."""
        ids = np.ascontiguousarray(ids, np.uint32)
        w = np.zeros(max(len(ids), 1)); vv = np.zeros(max(len(ids) * self.k, 1))
        L.check(L.lib().fmx_get_rows(self.h, _p(ids), C.c_int64(len(ids)), _p(w), _p(vv)))
        return w[: len(ids)], vv[: len(ids) * self.k].reshape(len(ids), self.k).T.copy()

    def set_rows(self, ids, w=None, v=None):
        ids = np.ascontiguousarray(ids, np.uint32)
        w = None if w is None else np.ascontiguousarray(w, np.float64)
        vv = None if v is None else np.ascontiguousarray(np.asarray(v, np.float64).T).ravel()
        L.check(L.lib().fmx_set_rows(self.h, _p(ids), C.c_int64(len(ids)), _p(w), _p(vv)))

    def save(self, path):
        L.check(L.lib().fmx_engine_save(self.h, str(path).encode()))

    def load(self, path):
        L.check(L.lib().fmx_engine_load(self.h, str(path).encode()))

    def predict(self, m, link=L.LINK_NONE):
        out = np.zeros(max(m.n, 1))
        L.check(L.lib().fmx_predict(self.h, m.h, _p(out), C.c_int(link)))
        return out[: m.n]

    def train(self, m, max_iter):
        done = C.c_int64()
        L.check(L.lib().fmx_train(self.h, m.h, C.c_int64(max_iter), C.byref(done)))
        return done.value

    def train_stream(self, total_rows, nnz_per_row=0, seed=0, row_offset=0, fields=None):
        """Streamed training on generated rows (fmx_train_stream); fields = (n_dense, field_vocab, skew) for the Criteo-shaped
        stream.  Returns (examples done, host seconds spent waiting for ingest)."""
        done, wait = C.c_int64(), C.c_double()
        spec = None
        if fields is not None:
            spec, keep = fields_spec(fields[0], fields[1], fields[2], seed)
        L.check(L.lib().fmx_train_stream(self.h, C.byref(spec) if spec is not None else None, C.c_int32(nnz_per_row), C.c_uint64(seed), C.c_int64(row_offset),
                                         C.c_int64(total_rows), C.byref(done), C.byref(wait)))
        return done.value, wait.value

    def train_tracked(self, m, max_iter, step_size, metric=L.EVAL_LL, convergence=1e-4, keep_params=True):
        """Learner::learn with the tracker on; returns dict(done, convergent, iters, evals[, params])."""
        tc = L.TrackConfig(C.sizeof(L.TrackConfig), int(metric), int(step_size), float(convergence), int(bool(keep_params)), 0)
        done, conv = C.c_int64(), C.c_int32()
        L.check(L.lib().fmx_train_tracked(self.h, m.h, C.c_int64(max_iter), C.byref(tc), C.byref(done), C.byref(conv)))
        n = C.c_int64()
        L.check(L.lib().fmx_trace_size(self.h, C.byref(n)))
        iters = np.zeros(max(n.value, 1), np.int64); evals = np.zeros(max(n.value, 1))
        L.check(L.lib().fmx_trace_get(self.h, _p(iters), _p(evals)))
        out = dict(done=done.value, convergent=bool(conv.value), iters=iters[: n.value], evals=evals[: n.value])
        if keep_params:
            out["params"] = [self.trace_params(i) for i in range(n.value)]
        return out

    def trace_params(self, record):
        w0 = C.c_double(); w = np.zeros(self.p); vv = np.zeros(max(self.k * self.p, 1))
        L.check(L.lib().fmx_trace_params(self.h, C.c_int64(record), C.byref(w0), _p(w), _p(vv)))
        return w0.value, w, vv[: self.k * self.p].reshape(self.p, self.k).T.copy()

    def evaluate(self, m, metric):
        out = C.c_double()
        L.check(L.lib().fmx_evaluate(self.h, m.h, C.c_int(metric), C.byref(out)))
        return out.value

    def train_order(self, m, order):
        order = np.ascontiguousarray(order, np.int64)
        L.check(L.lib().fmx_train_order(self.h, m.h, _p(order), C.c_int64(len(order))))

    def num_batches(self, m):
        nb = C.c_int64()
        L.check(L.lib().fmx_num_batches(self.h, m.h, C.byref(nb)))
        return nb.value

    def step(self, m, batch, rows_limit=0):
        L.check(L.lib().fmx_step(self.h, m.h, C.c_int64(batch), C.c_int64(rows_limit)))

    def grad(self, m, batch, rows_limit=0):
        L.check(L.lib().fmx_grad(self.h, m.h, C.c_int64(batch), C.c_int64(rows_limit)))

    def grad_buffer(self):
        ptr, n = C.c_void_p(), C.c_int64()
        L.check(L.lib().fmx_grad_buffer(self.h, C.byref(ptr), C.byref(n)))
        return ptr.value, n.value

    def grad_elem_bytes(self):
        b = C.c_int32()
        L.check(L.lib().fmx_grad_elem_bytes(self.h, C.byref(b)))
        return b.value

    def grad_layout(self):
        """(n_chunks, chunk_features, chunk_elems, tail_offset) of the exchange buffer."""
        a, b, c, d = C.c_int64(), C.c_int64(), C.c_int64(), C.c_int64()
        L.check(L.lib().fmx_grad_layout(self.h, C.byref(a), C.byref(b), C.byref(c), C.byref(d)))
        return a.value, b.value, c.value, d.value

    def grad_begin(self, m, batch, rows_limit=0):
        L.check(L.lib().fmx_grad_begin(self.h, m.h, C.c_int64(batch), C.c_int64(rows_limit)))

    def grad_chunk(self, m, chunk):
        L.check(L.lib().fmx_grad_chunk(self.h, m.h, C.c_int64(chunk)))

    def apply_chunk(self, chunk, global_rows=0, last=False):
        L.check(L.lib().fmx_apply_chunk(self.h, C.c_int64(chunk), C.c_int64(global_rows), C.c_int32(int(last))))

    def compact_info(self, m):
        """(record_elems, capacity, usable) of the compact exchange for this matrix (builds its tile plans)."""
        a, b, u = C.c_int64(), C.c_int64(), C.c_int32()
        L.check(L.lib().fmx_compact_info(self.h, m.h, C.byref(a), C.byref(b), C.byref(u)))
        return a.value, b.value, bool(u.value)

    def compact_count(self, m, batch):
        n = C.c_int64()
        L.check(L.lib().fmx_compact_count(self.h, m.h, C.c_int64(batch), C.byref(n)))
        return n.value

    def compact_reserve(self, capacity):
        L.check(L.lib().fmx_compact_reserve(self.h, C.c_int64(capacity)))

    def grad_compact(self, m, batch, rows_limit=0):
        L.check(L.lib().fmx_grad_compact(self.h, m.h, C.c_int64(batch), C.c_int64(rows_limit)))

    def compact_records(self):
        """(device pointer of the records, record count, device pointer of the 4-element tail)."""
        r, n, t = C.c_void_p(), C.c_int64(), C.c_void_p()
        L.check(L.lib().fmx_compact_records(self.h, C.byref(r), C.byref(n), C.byref(t)))
        return r.value, n.value, t.value

    def apply_compact(self, dev_records, counts, stride, global_rows=0):
        counts = np.ascontiguousarray(counts, np.int64)
        L.check(L.lib().fmx_apply_compact(self.h, C.c_void_p(dev_records), _p(counts), C.c_int32(len(counts)), C.c_int64(stride), C.c_int64(global_rows)))

    def apply(self, global_rows):
        L.check(L.lib().fmx_apply(self.h, C.c_int64(global_rows)))

    def apply_compact_parts(self, dev_records, counts, starts, global_rows=0):
        """parts at explicit record positions (what an uneven all-to-all leaves): fmx_apply_compact_parts"""
        counts = np.ascontiguousarray(counts, np.int64); starts = np.ascontiguousarray(starts, np.int64)
        L.check(L.lib().fmx_apply_compact_parts(self.h, C.c_void_p(dev_records), _p(counts), _p(starts), C.c_int32(len(counts)), C.c_int64(global_rows)))

    # owner-sharded exchange (include/fmx.h: feature j belongs to rank j mod N)
    def owner_configure(self, n_owners, rank):
        L.check(L.lib().fmx_owner_configure(self.h, C.c_int32(n_owners), C.c_int32(rank)))
        self._owners = n_owners

    def owner_info(self, m, batch=0):
        """(counts[n_owners], device pointer of the step's ids in owner-major order)"""
        counts = np.zeros(self._owners, np.int64)
        ids = C.c_void_p()
        L.check(L.lib().fmx_owner_info(self.h, m.h, C.c_int64(batch), _p(counts), C.byref(ids)))
        return counts, ids.value

    def rows_pack(self, dev_ids, n, dev_rows):
        re = C.c_int64()
        L.check(L.lib().fmx_rows_pack(self.h, C.c_void_p(dev_ids), C.c_int64(n), C.c_void_p(dev_rows), C.byref(re)))
        return re.value

    def rows_unpack(self, dev_ids, n, dev_rows):
        L.check(L.lib().fmx_rows_unpack(self.h, C.c_void_p(dev_ids), C.c_int64(n), C.c_void_p(dev_rows)))

    def source(self, total_rows, nnz_per_row=0, seed=0, row_offset=0, fields=None):
        """A streamed source of steps over generated rows (fmx_source_open): iterate with Source.next()."""
        return Source(self, total_rows, nnz_per_row, seed, row_offset, fields)

    def sync(self):
        L.check(L.lib().fmx_sync(self.h))

    def stream(self):
        s = C.c_void_p()
        L.check(L.lib().fmx_stream(self.h, C.byref(s)))
        return s.value

    def als_vsweep(self, m, error, alpha=1.0, v_lambda=None, v_mu=None, std_normals=None):
        """ALS V sweep; std_normals ((k, p) standard normal draws) switches to the MCMC (Gibbs) form."""
        error = np.ascontiguousarray(error, np.float64).copy()
        lam = None if v_lambda is None else np.ascontiguousarray(v_lambda, np.float64)
        mu = None if v_mu is None else np.ascontiguousarray(v_mu, np.float64)
        if std_normals is None:
            L.check(L.lib().fmx_als_vsweep(self.h, m.h, _p(error), C.c_double(alpha), _p(lam), _p(mu)))
        else:
            z = np.ascontiguousarray(std_normals, np.float64)
            if z.shape != (self.k, self.p):
                raise ValueError(f"std_normals must have shape ({self.k}, {self.p})")
            L.check(L.lib().fmx_mcmc_vsweep(self.h, m.h, _p(error), C.c_double(alpha), _p(lam), _p(mu), _p(z)))
        return error

    def vsweep_device(self, m, dev_error, alpha=1.0, v_lambda=None, v_mu=None, dev_std_normals=None):
        """ALS (dev_std_normals None) or MCMC V sweep on a residual that lives on the device (fmx_vsweep_device); pointers are ints."""
        lam = None if v_lambda is None else np.ascontiguousarray(v_lambda, np.float64)
        mu = None if v_mu is None else np.ascontiguousarray(v_mu, np.float64)
        L.check(L.lib().fmx_vsweep_device(self.h, m.h, C.c_void_p(dev_error), C.c_double(alpha), _p(lam), _p(mu),
                                          C.c_void_p(dev_std_normals) if dev_std_normals else None))

    def mcmc_train(self, m, max_iter, std_gammas, std_normals):
        """MCMC learner with caller-drawn variates (fmx.h: fmx_mcmc_train); returns (alpha, w_lambda, w_mu)."""
        g = np.ascontiguousarray(std_gammas, np.float64); z = np.ascontiguousarray(std_normals, np.float64)
        if g.shape != (max_iter, 2) or z.shape != (max_iter, 2 + self.p):
            raise ValueError(f"std_gammas must be ({max_iter}, 2) and std_normals ({max_iter}, {2 + self.p})")
        state = np.zeros(3)
        L.check(L.lib().fmx_mcmc_train(self.h, m.h, C.c_int32(max_iter), _p(g), _p(z), _p(state)))
        return tuple(state)

    def mcmc_train_from(self, m, max_iter, std_gammas, std_normals, state):
        """the chain continued from state = (alpha, w_lambda, w_mu); returns the new state"""
        g = np.ascontiguousarray(std_gammas, np.float64); z = np.ascontiguousarray(std_normals, np.float64)
        st = np.ascontiguousarray(state, np.float64).copy()
        L.check(L.lib().fmx_mcmc_train_from(self.h, m.h, C.c_int32(max_iter), _p(g), _p(z), _p(st)))
        return tuple(st)

    def mcmc_v_hyper(self, v_lambda, v_mu, std_gammas=None, std_normals=None):
        """update_v_lambda + update_v_mu; variates given: the MCMC draws, else the ALS means.  Returns (v_lambda, v_mu)."""
        lam = np.ascontiguousarray(v_lambda, np.float64).copy(); mu = np.ascontiguousarray(v_mu, np.float64).copy()
        sample = std_gammas is not None
        g = None if not sample else np.ascontiguousarray(std_gammas, np.float64)
        z = None if not sample else np.ascontiguousarray(std_normals, np.float64)
        L.check(L.lib().fmx_mcmc_v_hyper(self.h, _p(g), _p(z), _p(lam), _p(mu), C.c_int32(int(sample))))
        return lam, mu

    def als_plan(self, m):
        """(levels or groups, size of the largest, approximate?, level / group of every feature) of the ALS sweeps on this matrix."""
        lv, big, ap = C.c_int64(), C.c_int64(), C.c_int32()
        lof = np.zeros(max(self.p, 1), np.int32)
        L.check(L.lib().fmx_als_plan_info(self.h, m.h, C.byref(lv), C.byref(big), C.byref(ap), _p(lof)))
        return lv.value, big.value, bool(ap.value), lof[: self.p]

    def als_plan_kind(self, m):
        """0: the exact schedule (the reference's feature order); 1: the approximate groups (cfg.als_max_levels exceeded); 2: the coloured order (cfg.als_max_levels = -1, or -2 on a plan with long or heavy lists); 3: the coloured order nested feature-major (-2)."""
        ap = C.c_int32()
        L.check(L.lib().fmx_als_plan_info(self.h, m.h, None, None, C.byref(ap), None))
        return int(ap.value)

    def group_info(self):
        """a cfg.n_gpus handle: replicas, shared device or not, ordered device pairs / those with direct peer access, default exchange of sparse-tile steps"""
        v = [C.c_int32() for _ in range(5)]
        L.check(L.lib().fmx_group_info(self.h, *[C.byref(x) for x in v]))
        return {"replicas": v[0].value, "share_one_device": bool(v[1].value), "device_pairs": v[2].value, "device_pairs_with_direct_peer_access": v[3].value,
                "sparse_exchange": {0: "none", 1: "compact", 2: "owner"}.get(v[4].value, str(v[4].value))}

    def als_tiled(self, m):
        """(levels swept in the row-tiled form, rows per tile, tiles) for this matrix (fmx_als_tiled_info); (0, 0, 0): none."""
        lv, tr, nt = C.c_int32(), C.c_int64(), C.c_int32()
        L.check(L.lib().fmx_als_tiled_info(self.h, m.h, C.byref(lv), C.byref(tr), C.byref(nt)))
        return lv.value, tr.value, nt.value

    def als_level_order(self, m):
        """True when the V sweeps of this matrix take the level-order form (fmx_als_order_info: a complete tiled plan)."""
        v = C.c_int32()
        L.check(L.lib().fmx_als_order_info(self.h, m.h, C.byref(v)))
        return bool(v.value)

    def als_level_order_form(self, m):
        """0: the V sweeps go level by level through the three-pass / column-walking kernels; 1: the tile form of the level-order sweep; 2: the block form."""
        v = C.c_int32()
        L.check(L.lib().fmx_als_order_info(self.h, m.h, C.byref(v)))
        return int(v.value)

    @staticmethod
    def train_grid(engines, m, max_iter):
        """n reference-order learners side by side on one matrix and one visiting order (fmx_train_grid): every engine its own hyper-parameters and state."""
        arr = (C.c_void_p * len(engines))(*[e.h.value for e in engines])
        done = C.c_int64(0)
        L.check(L.lib().fmx_train_grid(arr, C.c_int32(len(engines)), m.h, C.c_int64(max_iter), C.byref(done)))
        return done.value

    def als_carry_q(self, on=True):
        """Opt-in: the block form and the feature-major form (als_max_levels = -2) of the V sweep keep q = X v_f current from sweep to sweep and skip the forward pass that rebuilds it (fmx_als_carry_q)."""
        L.check(L.lib().fmx_als_carry_q(self.h, C.c_int32(int(on))))

    def als_train(self, m, max_iter, with_v=False):
        L.check(L.lib().fmx_als_train(self.h, m.h, C.c_int32(max_iter), C.c_int32(int(with_v))))

    def profile(self, every=1):
        """every = 0: off; n > 0: HIP-event time every n-th launch of each kernel."""
        L.check(L.lib().fmx_profile_enable(self.h, C.c_int(int(every))))

    def profile_reset(self):
        L.check(L.lib().fmx_profile_reset(self.h))

    def profile_get(self, kernel):
        ms, n = C.c_double(), C.c_int64()
        L.check(L.lib().fmx_profile_get(self.h, C.c_int(kernel), C.byref(ms), C.byref(n)))
        return ms.value, n.value

    def w_in_row(self):
        """True when the fp32 tables use the w-in-row layout (fmx_layout_info)."""
        a, b = C.c_int32(), C.c_int32()
        L.check(L.lib().fmx_layout_info(self.h, C.byref(a), C.byref(b)))
        return bool(b.value)

    def rows_tune(self):
        """(serial, ms_serial, ms_pipelined): phase 1's schedule for large steps as this engine measured it (fmx_rows_tune_info)."""
        d, a, b = C.c_int32(), C.c_double(), C.c_double()
        L.check(L.lib().fmx_rows_tune_info(self.h, C.byref(d), C.byref(a), C.byref(b)))
        return d.value, a.value, b.value

    def close(self):
        if self.h:
            for src in list(getattr(self, "_sources", [])):   # their fmx_source points at this engine: close them while it exists
                try:
                    src.close()
                except Exception:
                    pass
            L.lib().fmx_engine_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass
