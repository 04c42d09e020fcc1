import os, sys, time
sys.path.insert(0, os.getcwd())
import numpy as np, torch
import rucene_amd
from rucene_amd import indexgen, _lib
docs = int(os.environ.get("DOCS", "100000000"))
seg = indexgen.build_zipf(docs, 1_000_000)
ctx = rucene_amd.Context()
sel = seg.terms[seg.terms["doc_freq"] >= 128]
total = int(sel["doc_freq"].sum())
td = torch.empty(total, dtype=torch.int32, device="cuda"); tf = torch.empty(total, dtype=torch.int32, device="cuda")
for rep in range(3):
    s2 = _lib.Segment(ctx, seg.doc_bytes, seg.norms, seg.max_doc, 0, None)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    s2.decode_terms_device(sel, td.data_ptr(), tf.data_ptr())
    torch.cuda.synchronize()
    print("rep", rep, "cold decode wall ms", round(1e3 * (time.perf_counter() - t0), 3), "terms", sel.size, flush=True)
    t0 = time.perf_counter()
    s2.decode_terms_device(sel, td.data_ptr(), tf.data_ptr())
    torch.cuda.synchronize()
    print("rep", rep, "warm decode wall ms", round(1e3 * (time.perf_counter() - t0), 3), flush=True)
    s2.close()
ctx.close()
