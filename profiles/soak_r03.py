"""Round-3 soak at configs[3]'s shape (33 M features, k = 32, 262 144-row steps, streamed): 150 steps, run in separate processes with the per-field sort /
the pair sort, the long lists beside / behind the main phase-2 kernel, planning on a second stream / on the engine's stream, one GPU / two replicas behind one handle
(owner-sharded and all-gather exchange, both on this device) -- every single-GPU variant must leave the same bits in the rows of the features that occurred, and so
must the two exchanges of the two-replica job.  python profiles/soak_r03.py"""
import hashlib, os, subprocess, sys
CHILD = r'''
import hashlib, sys
import numpy as np
sys.path.insert(0, ".")
from fmwr_amd import _lib as L, engine
n_gpus, steps = int(sys.argv[1]), int(sys.argv[2])
vocab, B = engine.CRITEO_VOCAB, 262_144
p = 13 + sum(vocab)
first = engine.Matrix.synthetic_fields(60_000, 13, vocab, 3.0, 77)
touched = np.unique(first.export()[1]); first.close()
kw = dict(task=L.TASK_CLASSIFICATION, solver=L.SOLVER_SGD, num_factor=32, learn_rate=0.05, l2_w1=1e-4, l2_v=1e-4, mode=L.MODE_MINIBATCH, batch_rows=B)
if n_gpus > 1:
    kw.update(n_gpus=n_gpus, gpus_share_device=1)
e = engine.Engine(p, **kw)
e.init_normal(20240001, 0.0, 0.01)
done, _ = e.train_stream(steps * B * n_gpus, seed=77, fields=(13, vocab, 3.0))
assert done == steps * B * n_gpus
w, v = e.get_rows(touched)
assert np.all(np.isfinite(v)) and np.all(np.isfinite(w))
print("HASH", hashlib.sha256(v.tobytes() + w.tobytes()).hexdigest()[:16])
'''
bad = 0
for n_gpus, steps, variants in ((1, 150, (("default", {}), ("pair sort", {"FMX_FIELD_SORT": "0"}), ("general sort", {"FMX_FIELDS_SPLIT": "0"}), ("long lists behind", {"FMX_LONG_SIDE": "0"}),
                                         ("one ingest stream", {"FMX_STREAM_OVERLAP": "0"}), ("default again", {}))),
                                (2, 40, (("owner", {"FMX_GROUP_EXCHANGE": "owner"}), ("compact", {"FMX_GROUP_EXCHANGE": "compact"}), ("owner again", {"FMX_GROUP_EXCHANGE": "owner"})))):
    out = {}
    for name, env in variants:
        r = subprocess.run([sys.executable, "-c", CHILD, str(n_gpus), str(steps)], env=dict(os.environ, **env), capture_output=True, text=True, timeout=900)
        line = [l for l in r.stdout.splitlines() if l.startswith("HASH")]
        out[name] = line[0].split()[1] if line else "FAILED " + r.stderr[-400:]
    same = len(set(out.values())) == 1
    bad += not same
    print(f"n_gpus {n_gpus}, {steps} streamed steps: {'identical' if same else 'DIFFERENT'} {out}", flush=True)
sys.exit(1 if bad else 0)
