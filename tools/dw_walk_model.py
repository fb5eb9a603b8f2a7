"""Model of the HBM operand fetch of a grouped weight-gradient launch (lav_gemm_tn_grouped) under a given block walk: every XCD fetches each
distinct (job, K-slab, operand panel) its resident blocks touch once (blocks of a launch <= 256 are co-resident and advance through K together).
Prints fetched / algorithmic panel-slab counts for the round-4 walk (per-job XCD runs, n-fastest) and the round-5 walk (launch-wide XCD runs,
short dimension fastest)."""
import sys


def walk(jobs, splits, mode):
    blocks = []   # (job, split, tm, tn) in launch order
    for j, (M, N) in enumerate(jobs):
        tm_, tn_ = (M + 255) // 256, N // 256
        blocks.append((j, tm_, tn_))
    tot = sum(tm * tn * splits for _, tm, tn in blocks)
    per_xcd = [set() for _ in range(8)]
    def remap(b, tot):
        q, r, xcd, idx = tot >> 3, tot & 7, b & 7, b >> 3
        return (xcd * (q + 1) if xcd < r else r * (q + 1) + (xcd - r) * q) + idx
    blk0 = []
    o = 0
    for j, tm, tn in blocks:
        blk0.append(o); o += tm * tn * splits
    for b in range(tot):
        xcd = b & 7
        bb = remap(b, tot) if mode else b
        j = max(i for i in range(len(blocks)) if blk0[i] <= bb)
        _, tm, tn = blocks[j]
        bid = bb - blk0[j]
        if not mode:
            bid = remap(bid, tm * tn * splits)   # NOTE: hardware xcd = b & 7 where b = blk0 + bid_in; blk0 % 8 may be != 0 (relabel only)
        sp, t = divmod(bid, tm * tn)
        if mode and tm < tn:
            n_i, m_i = divmod(t, tm)
        else:
            m_i, n_i = divmod(t, tn)
        per_xcd[xcd].add((j, sp, 'A', m_i)); per_xcd[xcd].add((j, sp, 'B', n_i))
    fetched = sum(len(s) for s in per_xcd)
    alg = sum((tm + tn) * splits for _, tm, tn in blocks)
    return tot, fetched, alg


if __name__ == "__main__":
    groups = {"fusion layer (K = 45120)": ([(768, 3072), (3072, 768), (768, 768), (2304, 768)], 2),
              "swin stage 2 (K = 31360)": ([(512, 2048), (2048, 512), (512, 512), (1536, 512)], 4),
              "swin stage 3 (K = 7840)": ([(1024, 4096), (4096, 1024), (1024, 1024), (3072, 1024)], 1)}
    for name, (jobs, sp) in groups.items():
        for mode in (0, 1):
            tot, f, a = walk(jobs, sp, mode)
            print(f"{name}: walk {mode}: {tot} blocks, panel-slabs fetched {f} vs algorithmic {a} = {f / a:.2f}x")
