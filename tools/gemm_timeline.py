"""Per-workgroup phase timeline of the 128x128 MFMA GEMM kernel (measurement build, see tools/experiments/build_timeline_lib.sh):
  OFASYS_AMD_LIB=tools/experiments/_build/libofasys_amd_tl.so OFA_GEMM_TILE=22 OFA_GEMM_SPLIT_MIN_K=1000000 \
      python tools/gemm_timeline.py [kind M N K ...]
Thread 0 of every workgroup stamps s_memtime at entry / first tile landed / K loop done / epilogue issued / stores acknowledged
and after every K-step, plus HW_ID and XCC_ID; this script groups the workgroups by compute unit and prints where a tile's
life goes and what a CU slot does between two tiles."""
import os
import sys
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ofasys_amd import kernels as K

dev = 'cuda'
torch.manual_seed(0)
argv = sys.argv[1:]
shapes = [(argv[i], int(argv[i + 1]), int(argv[i + 2]), int(argv[i + 3])) for i in range(0, len(argv), 4)] or \
    [("NT", 13312, 3072, 768), ("NT", 13312, 2304, 768), ("NN", 13312, 768, 3072)]
ws = K.workspace(256 << 20, torch.device(dev, torch.cuda.current_device()), "gemm")
for kind, m, n, k in shapes:
    ta, tb = {'NT': (False, True), 'NN': (False, False), 'TN': (True, False)}[kind]
    a = torch.randn((k, m) if ta else (m, k), device=dev).bfloat16()
    b = torch.randn((n, k) if tb else (k, n), device=dev).bfloat16()
    out = torch.empty(m, n, device=dev, dtype=torch.bfloat16)
    for _ in range(3):
        K.gemm(a, b, ta, tb, out=out)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10):
        K.gemm(a, b, ta, tb, out=out)
    e1.record()
    torch.cuda.synchronize()
    us = e0.elapsed_time(e1) * 100
    ws.zero_()
    K.gemm(a, b, ta, tb, out=out)
    torch.cuda.synchronize()
    bm, bn = {"42": (256, 128), "84": (256, 256), "44": (256, 256), "85": (256, 256)}.get(os.environ.get("OFA_GEMM_TILE", "22"), (128, 128))
    ntiles = ((m + bm - 1) // bm) * ((n + bn - 1) // bn)
    if os.environ.get("OFA_GEMM_TILE") == "85":      # persistent: one record per workgroup
        ntiles = min(256, (ntiles + 7) // 8 * 8)
    tl = ws.view(torch.int64)[:ntiles * 32].view(ntiles, 32).cpu().numpy()
    if (tl[:, 0] == 0).any():
        print(f"{kind} {m} {n} {k}: {int((tl[:, 0] == 0).sum())} workgroups without a stamp (not the 128x128 kernel?)")
        continue
    nk = k // 64
    hw, xcc = tl[:, 5], tl[:, 6] & 0xf
    cu, se, sh = (hw >> 8) & 0xf, (hw >> 13) & 0x7, (hw >> 12) & 1
    cuid = ((xcc * 8 + se) * 2 + sh) * 16 + cu
    # shader clocks per 100 MHz tick (calibration over the whole launch)
    # (s_memtime is per XCC: calibrate on each workgroup's own life, compare stamps only inside one XCC)
    span_rt = (tl[:, 31].max() - tl[:, 7].min()) / 100.0
    mhz = float(np.median((tl[:, 4] - tl[:, 0]) / np.maximum(tl[:, 31] - tl[:, 7], 1))) * 100.0
    c2us = 1.0 / mhz
    if os.environ.get("OFA_TL_DUMP"):
        np.save(os.path.join(os.environ["OFA_TL_DUMP"], f"tl_{kind}_{m}_{n}_{k}.npy"), tl)
    pro, main, epi, ack = (tl[:, 1] - tl[:, 0]) * c2us, (tl[:, 2] - tl[:, 1]) * c2us, (tl[:, 3] - tl[:, 2]) * c2us, \
        (tl[:, 4] - tl[:, 3]) * c2us
    life = (tl[:, 4] - tl[:, 0]) * c2us
    ksteps = np.diff(tl[:, 8:8 + min(nk, 14)], axis=1) * c2us
    print(f"== {kind} {m} {n} {k}: {us:.1f} us/launch (10 back to back); probe launch spans {span_rt:.1f} us, s_memtime = "
          f"{mhz:.0f} MHz; {ntiles} workgroups on {len(np.unique(cuid))} CUs")

    def q(x):
        return f"mean {x.mean():6.2f}  p10 {np.percentile(x, 10):6.2f}  p50 {np.percentile(x, 50):6.2f}  p90 {np.percentile(x, 90):6.2f} us"
    print(f"  entry -> first tile landed : {q(pro)}")
    print(f"  K loop ({nk} steps)        : {q(main)}")
    print(f"    per K-step (steps 2..)  : {q(ksteps.reshape(-1))}")
    print(f"  epilogue (issue)          : {q(epi)}")
    print(f"  store acknowledgement     : {q(ack)}")
    print(f"  workgroup life            : {q(life)}")
    if tl[:, 24].any():   # ring kernel: stamps inside K-step 6 of wave 0 (shader clocks)
        f = tl[:, 24:31].astype(np.float64)
        d = np.diff(f, axis=1)
        names = ["barrier -> slice-0 fragments in", "slice 0: 4 MFMA + DMA quarter -> slice-1 frags in", "slice 1 -> slice-2 frags in",
                 "slice 2 -> slice-3 frags in", "slice 3: 4 MFMA + DMA quarter issued", "end of step -> through next barrier"]
        for i, nm in enumerate(names):
            print(f"    K-step 6, {nm:52s}: p10 {np.percentile(d[:, i], 10):6.0f}  p50 {np.percentile(d[:, i], 50):6.0f}  p90 {np.percentile(d[:, i], 90):6.0f} clk")
    # per CU: order by entry; how many live at once, what a freed slot waits for
    rt_first = tl[:, 7].min()
    slots = int(os.environ.get("OFA_TL_SLOTS", "2"))
    gaps, first_entry, conc = [], [], []
    for c in np.unique(cuid):
        idx = np.where(cuid == c)[0]
        idx = idx[np.argsort(tl[idx, 0])]
        first_entry.append((tl[idx[0], 7] - rt_first) / 100.0)
        ends = np.sort(tl[idx, 4])
        starts = tl[idx, 0]
        # the i-th start beyond the first `slots` takes the slot of the (i-slots)-th end
        for i in range(slots, len(idx)):
            gaps.append((starts[i] - ends[i - slots]) * c2us)
        conc.append(len(idx))
    gaps = np.array(gaps) if gaps else np.zeros(1)
    print(f"  workgroups per CU         : min {min(conc)} max {max(conc)}; first entry after launch: {q(np.array(first_entry))}")
    print(f"  slot freed -> next entry  : {q(gaps)}")
    busy = (main.sum()) / (len(np.unique(cuid)) * slots * span_rt)
    print(f"  share of ({slots} slots x span) spent inside K loops: {busy:.2f}")
    # one CU in detail
    c = np.unique(cuid)[len(np.unique(cuid)) // 2]
    idx = np.where(cuid == c)[0]
    idx = idx[np.argsort(tl[idx, 0])]
    print(f"  CU {c} (xcc {c >> 8}):  entry  landed  loop-end  epi-issued  acked   [us from the XCC's first entry]")
    t_first = tl[xcc == (c >> 8), 0].min()
    for i in idx:
        r = (tl[i, :5] - t_first) * c2us
        print(f"      wg {i:5d}: {r[0]:7.2f} {r[1]:7.2f} {r[2]:8.2f} {r[3]:9.2f} {r[4]:8.2f}")
