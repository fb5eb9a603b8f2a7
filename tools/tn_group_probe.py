"""GPU probe: the four weight-gradient GEMMs of a Swin-B stage-2 block / of a fusion layer launched one by one (their own split factors)
vs ONE grouped launch (lav_gemm_tn_grouped) at several group split factors."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from lavender_amd import hip as K
from tools._bench import bench

GROUPS = {
    "Swin-B stage 2 block (31360 rows)": [(512, 2048, 31360), (2048, 512, 31360), (1536, 512, 31360), (512, 512, 31360)],
    "Swin-B stage 3 block (7840 rows)": [(1024, 4096, 7840), (4096, 1024, 7840), (3072, 1024, 7840), (1024, 1024, 7840)],
    "fusion layer (45120 rows)": [(768, 3072, 45120), (3072, 768, 45120), (2304, 768, 45120), (768, 768, 45120)],
}
for name, shapes in GROUPS.items():
    ops = []
    for (M, N, Kd) in shapes:
        A = torch.randn(Kd, M, device="cuda").bfloat16(); B = torch.randn(Kd, N, device="cuda").bfloat16()
        ops.append((M, N, Kd, A, B, torch.zeros(M, N, device="cuda")))
    tiles = [((M + 255) // 256) * (N // 256) for (M, N, Kd, *_ ) in ops]
    cur = [K.splits_for(M, N, Kd) for (M, N, Kd, *_ ) in ops]
    def separate():
        for o, s in zip(ops, cur):
            K.gemm(2, o[3], o[4], o[0], o[1], o[2], out=o[5], accumulate=True, splits=s)
    t_sep = bench(separate)
    line = f"{name}: tiles {tiles}, separate (splits {cur}) {t_sep:.0f} us;  grouped:"
    jobs = [dict(A=o[3], B=o[4], out=o[5], fallback_splits=s) for o, s in zip(ops, cur)]
    for gs in (1, 2, 3, 4, 5):
        if sum(tiles) * gs > 320:
            continue
        line += f"  s={gs} ({sum(tiles) * gs} blocks) {bench(lambda: K.gemm_tn_grouped(jobs, gs)):.0f} us"
    print(line, flush=True)
