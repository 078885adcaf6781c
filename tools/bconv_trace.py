"""Occupancy trace of one bf16 conv launch (diagnostic build with -DCNB_TRACE; see cn_bconv.hip): start / end of EVERY
block on the 100 MHz real-time counter + its HW_ID, i.e. how many blocks each CU really holds over the launch and how
long a slot stays empty between two blocks.

    make -C cultionet_amd/csrc trace     # builds libcultionet_hip_trace.so
    CN_LIB_PATH=cultionet_amd/csrc/libcultionet_hip_trace.so python tools/bconv_trace.py 32 128 100 100 128 3
"""
import collections
import ctypes
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from cultionet_amd import _lib

B, Cin, H, W, Cout, k = (int(v) for v in sys.argv[1:7])
dev = torch.device("cuda:0")
BF = torch.bfloat16
T, p = k * k, k // 2
s = torch.cuda.current_stream().cuda_stream
x = torch.randn(B, H, W, Cin, device=dev).to(BF)
w = torch.randn(Cout, Cin, k, k, device=dev) * (Cin * T) ** -0.5
wp = torch.empty(_lib.query("cn_bconv_packed_elems", T, Cin, Cout), dtype=BF, device=dev)
_lib.call("cn_pack_weights_bf16", w.data_ptr(), wp.data_ptr(), T, Cin, Cout, T, Cin * T, 1, s)
y = torch.empty(B, H, W, Cout, dtype=BF, device=dev)
rows = _lib.query("cn_conv2d_stats_rows_bf16", B, H, W, Cout, k, k, 1, p, 1)
stats = torch.empty(rows * 2 * Cout, device=dev)
for _ in range(5):
    _lib.call("cn_conv2d_fwd_bf16", x.data_ptr(), Cin, wp.data_ptr(), None, y.data_ptr(), Cout, 0, B, Cin, H, W, Cout, k, k,
              1, p, 1, 0, 0, stats.data_ptr(), s)
torch.cuda.synchronize()
lib = _lib.load()
n = min(rows * ((Cout + 127) // 128), 16384)
buf = (ctypes.c_ulonglong * (3 * n))()
lib.cn_bconv_read_trace.argtypes = [ctypes.c_void_p, ctypes.c_int]
assert lib.cn_bconv_read_trace(buf, n) == 0
blocks = [(buf[3 * i], buf[3 * i + 1], buf[3 * i + 2]) for i in range(n) if buf[3 * i + 1]]
t0 = min(b[0] for b in blocks)
t1 = max(b[1] for b in blocks)
TICK = 10.0  # ns per s_memrealtime tick
print(f"{len(blocks)} blocks, launch span {(t1 - t0) * TICK / 1e3:.1f} us")
life = sorted((b[1] - b[0]) * TICK / 1e3 for b in blocks)
print(f"block life us: min {life[0]:.1f}  p10 {life[len(life) // 10]:.1f}  median {life[len(life) // 2]:.1f}  "
      f"p90 {life[len(life) * 9 // 10]:.1f}  max {life[-1]:.1f}   sum / span = {sum(life) / ((t1 - t0) * TICK / 1e3):.1f} "
      f"blocks resident on average")
# HW_ID (gfx9): wave_id[3:0] simd_id[5:4] pipe[7:6] cu_id[11:8] sh_id[12] se_id[15:13] ... (xcc in XCC_ID, not here)
per_cu = collections.defaultdict(list)
for a, b, hw in blocks:
    per_cu[(hw >> 8) & 0xFF].append((a, b))
print(f"distinct (se, sh, cu) ids seen: {len(per_cu)} (XCDs alias: 8 XCDs share each id)")
# concurrency histogram over time (whole chip)
ev = sorted([(a, 1) for a, _, _ in blocks] + [(b, -1) for _, b, _ in blocks])
cur, last, hist = 0, t0, collections.Counter()
for t, d in ev:
    hist[cur] += t - last
    cur += d
    last = t
tot = sum(hist.values())
cum = 0
print("chip-wide resident blocks (time share):")
for lo in range(0, 800, 96):
    share = sum(v for c, v in hist.items() if lo <= c < lo + 96) / tot
    print(f"   {lo:4d}-{lo + 95:4d}: {share:6.1%}")
# start-time distribution: are blocks launched in waves?
starts = sorted((a - t0) * TICK / 1e3 for a, _, _ in blocks)
print("block starts per 10 us:", [sum(1 for v in starts if lo <= v < lo + 10) for lo in range(0, int(starts[-1]) + 10, 10)])
