#!/usr/bin/env python3
"""
Which launch of the published network is not repeatable?  (r04: how the store-data hazard behind the lean epilogue was
found, profiles/r04_store_data_hazard_plain.txt.)  Replays the Python plan step by step; every conv step is re-run REP
times on restored buffers and compared bitwise with its first result and with the same descriptor through a second,
trusted build of the library loaded beside the one under test:

    DDPM3D_LIB=<build under test> DDPM3D_TRUSTED_LIB=<trusted build> python tools/replay_repeatability.py bf16 8,32,32
"""
import ctypes as C, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (os.path.join(ROOT, "3d-denoising-diffusion-model_amd"), ROOT):
    sys.path.insert(0, p)
import numpy as np, torch, bench
from guided_diffusion import synth, _hip as H
prec = sys.argv[1] if len(sys.argv) > 1 else "bf16"
S = [int(v) for v in (sys.argv[2] if len(sys.argv) > 2 else "8,32,32").split(",")]
REP = 12
hip = C.CDLL("libamdhip64.so")
TRUST = C.CDLL(os.path.join(ROOT, os.environ.get("DDPM3D_TRUSTED_LIB", "scratch/pwold/libddpm3d.so")), mode=os.RTLD_LOCAL)
TRUST.ddpm3d_conv3d.argtypes = [C.c_void_p, C.c_void_p]
TRUST.ddpm3d_conv3d.restype = C.c_int
dev = torch.device("cuda:0")
model, _, _ = bench.build_model(bench.PUBLISHED, "250", dev)
model.conv_precision = prec
shape = (1, 1, *S)
x = torch.from_numpy(synth.synth_noise(shape, 1, seed=3)[0]).to(dev)
lr = torch.from_numpy(synth.synth_low_res(shape, seed=1234)).to(dev)
t = torch.full((1,), 251, dtype=torch.long, device=dev)
eng = model.engine()
rows = eng.film_rows(t.float())
with torch.no_grad():
    eng.forward(x, lr, rows, eng.film_total)
torch.cuda.synchronize()
plan = eng.plan(1, *S)
# patch per-call pointers the way _enqueue does, then walk the steps by hand
out_buf = plan.out_buf
plan._enqueue(x.data_ptr(), lr.data_ptr(), rows.data_ptr(), eng.film_total, out_buf.data_ptr())
torch.cuda.synchronize()
st = H.stream()
def dcopy(dst, src, n):
    assert hip.hipMemcpy(C.c_void_p(dst), C.c_void_p(src), C.c_size_t(n), 3) == 0
bad = 0
for i, (fn, args) in enumerate(plan.steps):
    args[-1] = st
    meta = plan.conv_meta.get(i)
    if meta is None or meta[0] == "pool_act" or meta[0].startswith("attention"):
        assert fn(*args) == 0
        continue
    d = args[0]._obj
    es = 2 if (d.io_dtype & H.IO_OUT_BF16) else 4
    n = d.N * d.D * d.H * d.W * d.Cout * es
    if d.out_layout != H.OUT_NDHWC:
        n = d.N * d.D * d.H * d.W * d.Cout * 4
    before = torch.empty(n, dtype=torch.uint8, device=dev)
    first = torch.empty(n, dtype=torch.uint8, device=dev)
    cur = torch.empty(n, dtype=torch.uint8, device=dev)
    torch.cuda.synchronize()
    dcopy(before.data_ptr(), d.out, n)
    nst = 0
    sfirst = scur = None
    if d.stats:
        nst = d.N * d.Cout * d.stats_rows * 16
        sfirst = torch.empty(nst, dtype=torch.uint8, device=dev); scur = torch.empty(nst, dtype=torch.uint8, device=dev)
    assert fn(*args) == 0
    torch.cuda.synchronize()
    dcopy(first.data_ptr(), d.out, n)
    if nst: dcopy(sfirst.data_ptr(), d.stats, nst)
    # the same descriptor through the trusted build (pre-lean library)
    dcopy(d.out, before.data_ptr(), n)
    assert TRUST.ddpm3d_conv3d(C.addressof(d), st) == 0
    torch.cuda.synchronize()
    dcopy(cur.data_ptr(), d.out, n)
    tdiff = int((cur != first).sum())
    if tdiff:
        print("   step %d %s: FIRST run differs from the trusted library's result in %d bytes" % (i, meta[0], tdiff))
    trusted = cur.clone()
    ndiff = nsd = 0
    results = {}
    for r in range(REP):
        dcopy(d.out, before.data_ptr(), n)
        assert fn(*args) == 0
        torch.cuda.synchronize()
        dcopy(cur.data_ptr(), d.out, n)
        key = (int((cur != trusted).sum()), int((cur != first).sum()))
        results[key] = results.get(key, 0) + 1
        df = (cur != first)
        if df.any():
            ndiff += 1
            if ndiff == 1:
                idx = torch.nonzero(df).flatten()[:12].tolist()
                el = [j // es for j in idx]
                print("   step %d %s: %d differing bytes; first elements %s (cout %s, voxel %s)" % (
                    i, meta[0], int(df.sum()), el, [e % d.Cout for e in el], [e // d.Cout for e in el]))
                dt = torch.bfloat16 if es == 2 else torch.float32
                fa, ca = first.view(dt).float(), cur.view(dt).float()
                for e in sorted(set(el))[:6]:
                    c, v = e % d.Cout, e // d.Cout
                    print("      elem %d (cout %d vox %d): first %.6g now %.6g | same cout, voxels +4..+32 step 4 (first): %s" % (
                        e, c, v, fa[e].item(), ca[e].item(), [round(fa[(v + k) * d.Cout + c].item(), 5) for k in range(4, 36, 4) if (v + k) * d.Cout + c < fa.numel()]))
        if nst:
            dcopy(scur.data_ptr(), d.stats, nst)
            if (scur != sfirst).any(): nsd += 1
    dcopy(d.out, first.data_ptr(), n)
    if nst: dcopy(d.stats, sfirst.data_ptr(), nst)
    tag = "%3d %-24s Cin %4d Cout %4d %dx%dx%d split? ws=%d res_mode=%d io=%d" % (i, meta[0], d.Cin, d.Cout, d.D, d.H, d.W, d.workspace_bytes, d.res_mode, d.io_dtype)
    if ndiff or nsd or tdiff:
        print("   reruns by (bytes differing from trusted, from first run): %s" % results)
    if ndiff or nsd:
        bad += 1
        print("NOT REPEATABLE:", tag, "runs differing: out %d stats %d of %d" % (ndiff, nsd, REP))
print("done, %d non-repeatable conv steps" % bad)
