"""Do global stores issued beside the LDS-DMA stream cost the persistent GEMM's K loop anything?  (round 6, the question behind an
epilogue inside the K loop.)  Tuning library only: VF_G8_DBG bit 16 / 32 adds one / two 16-byte-per-lane stores per phase to
gemm8x_kernel (32 / 64 KiB per K-tile and block, against 64 KiB of LDS-DMA fill and 128 KiB per tile of real epilogue stores at
the end); bits 64 / 128 = the pattern of an epilogue inside the K loop: 4 stores per wave in P1 and P2 of a tile's first K-tile (64),
in P3 and P4 of its second (128): 64 KiB per tile each, on top of the real epilogue; the counted waits leave them in flight for
1.75 / 1.5 and 1.25 / 1 K-tiles.  Interleaved rounds in one process; results of the probe launches are meaningless."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from variantformer_amd import ops, _lib
from variantformer_amd.csrc.build import TUNING_LIB
_lib.load(TUNING_LIB)
SHAPES = [("s2r Wqkv  K=512", 769460, 1536, 512, ops.EPI_BF16), ("s2r geglu K=512", 769460, 2048, 512, ops.EPI_GEGLU_BF16),
          ("gene Wqkv K=1536", 86832, 4608, 1536, ops.EPI_BF16), ("gene geglu K=1536", 86832, 2048, 1536, ops.EPI_GEGLU_BF16)]
MODES = [0, 64, 128, 192]
for name, M, N, K, epi in SHAPES:
    a = (torch.rand((M, K), device="cuda") * 2 - 1).bfloat16()
    w = ((torch.rand((N, K), device="cuda") * 2 - 1) / K ** 0.5).bfloat16()
    b = torch.rand((N,), device="cuda")
    times = {m: [] for m in MODES}
    for rnd in range(7):
        for m in MODES:
            os.environ["VF_G8_DBG"] = str(m)
            s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            s.record()
            ops.gemm(a, w, b, epi, variant=22)
            e.record()
            torch.cuda.synchronize()
            if rnd:
                times[m].append(s.elapsed_time(e))
    os.environ["VF_G8_DBG"] = "0"
    fl = 2.0 * M * N * K
    tiles = ((M + 255) // 256) * ((N + 255) // 256)
    kt = K // 64
    line = []
    for m in MODES:
        t = sorted(times[m])
        extra_kib = ({0: 0, 16: 32, 32: 64}.get(m, 0) * kt + {64: 64, 128: 64, 192: 128}.get(m, 0)) * tiles
        line.append(f"dbg={m:2d}: min {t[0] * 1e3:8.1f} us median {t[len(t) // 2] * 1e3:8.1f} us ({fl / t[0] / 1e9:6.0f} TFLOP/s; +{extra_kib / 2 ** 20:6.2f} GiB of probe stores)")
    print(f"{name:18s} " + " | ".join(line), flush=True)
