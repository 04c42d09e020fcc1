import os, sys
import numpy as np
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import torch, rucene_amd
from rucene_amd import indexgen, _lib as gpu
max_doc = 700_000
rng = np.random.default_rng(606)
blocks = [15, 16, 17, 63, 64, 65, 127, 128, 129, 511, 512, 513, 1023, 1024, 1025, 2047, 2048, 4095, 4096, 4200]
lists = []
for i, nb in enumerate(blocks):
    df = 128 * nb + (0 if i % 3 == 0 else int(rng.integers(1, 128)))
    docs = np.sort(rng.permutation(max_doc)[:df]).astype(np.int32)
    freqs = rng.integers(1, 4, size=df).astype(np.int32)
    freqs[rng.integers(0, df, size=5)] = 9
    lists.append((docs, freqs))
norms = rng.choice(np.array([100, 110, 124], dtype=np.uint8), size=max_doc)
seg = indexgen.build_explicit(max_doc, lists, norms=norms)
dfs = np.array([l[0].size for l in lists])
ctx = rucene_amd.Context()
leaf = rucene_amd.LeafReader(seg.doc_bytes, seg.norms, max_doc, seg.terms, sum_total_term_freq=30 * max_doc)
g = rucene_amd.GpuIndexSearcher([leaf], ctx=ctx)
"""Stress of the in-kernel fold (TermMerge): batches of DIFFERENT item layouts alternate, so that a list or count read stale — left in
an XCD's L2 or in memory by the launch before — shows as a wrong total or row (the same batch over and over hides it: stale = fresh)."""
batches = [np.arange(20)[::-1].copy(), np.arange(12, 14), np.arange(20), np.arange(4, 5), np.array([19, 0, 18, 1, 17, 2, 6, 6, 6, 9]), np.arange(6, 20)]
batches = [b.astype(np.int64).reshape(-1, 1) for b in batches]
packed = [g.pack_uniform(gpu.OP_TERM, b, leaf) for b in batches]
K = 10
refs = {}
bad = {"two": 0, "fused": 0}
n = 0
rng2 = np.random.default_rng(1)
for it in range(int(os.environ.get("ITERS", "400"))):
    bi = int(rng2.integers(0, len(batches)))
    ids = batches[bi]
    for mode in ("two", "fused"):
        nq = ids.shape[0]
        hits = torch.full((nq, K), -3, dtype=torch.int64, device="cuda")
        totals = torch.full((nq,), -3, dtype=torch.int64, device="cuda")
        if mode == "fused":
            g.search_uniform_device(gpu.OP_TERM, ids, leaf, K, hits.data_ptr(), totals.data_ptr())
        else:
            leaf.segment.search_batch_device(packed[bi][0], packed[bi][1], K, hits.data_ptr(), totals.data_ptr())
        ctx.synchronize()
        t = totals.cpu().numpy(); h = hits.cpu().numpy()
        n += 1
        if bi not in refs: refs[bi] = h.copy()
        if not (t == dfs[ids[:, 0]]).all() or not (h == refs[bi]).all():
            bad[mode] += 1
            if bad[mode] <= 3: print(mode, it, "batch", bi, "totals off at", np.nonzero(t != dfs[ids[:, 0]])[0], (t - dfs[ids[:, 0]])[t != dfs[ids[:, 0]]], "rows differ", int((h != refs[bi]).any(axis=1).sum()))
print("FOLD", os.environ.get("RGPU_TERM_FOLD", "1"), "bad launches of", n, ":", bad)
