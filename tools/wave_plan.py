#!/usr/bin/env python3
"""The far (bulk-stream) launches of one batch call, restated from gs_wave_bulk_plan / gs_lml_wave (csrc/host/wave.hip.h): for every
k_gemm_ld3g launch its members, M, K, algorithmic flops and algorithmic HBM bytes.
   algorithmic bytes of a member = its C lower triangle read and written once (16 B per element of M (M + 1) / 2) + its M x K panel
   rows read once (both operands of the symmetric product are those rows)."""
GS_BORDER = 16


def bulk_plan(np_, depth, deep_min_rows, first_len):
    S = np_ // 256
    plan = [None] * S
    a = 0
    while a < S:
        r2 = 256 * (a + 1)
        L = 1
        while L < depth and r2 + 256 * (L + 1) <= np_:
            L += 1
        if L > 2 and np_ + GS_BORDER - r2 < deep_min_rows:
            L = 2
        if a == 0 and first_len > 0:
            L = min(L, first_len)
        for i in range(L):
            plan[a + i] = {"near": i < L - 1, "K": 256 * (i + 1), "first": a}
        a += L
    return plan


def far_launches(n, group_sizes, depth=4, deep_min_rows=3072, head=(1, 2, 4)):
    np_ = (n + 255) // 256 * 256
    naug = np_ + GS_BORDER
    out = []
    for g, cnt in enumerate(group_sizes):
        plan = bulk_plan(np_, depth, deep_min_rows, head[g] if g < len(head) else 0)
        for s, st in enumerate(plan):
            if st["near"]:
                continue
            M = naug - 256 * (s + 1)
            if M <= 0:
                continue
            K = st["K"]
            out.append({"group": g, "step": s, "members": cnt, "M": M, "K": K, "flops": cnt * M * (M + 1) * K,
                        "bytes": cnt * (16 * M * (M + 1) // 2 + 8 * M * K)})
    return out


if __name__ == "__main__":
    import json
    import sys
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 8192
    sizes = [int(v) for v in sys.argv[2].split(",")] if len(sys.argv) > 2 else [7, 7, 6]
    L = far_launches(n, sizes)
    print(json.dumps({"launches": len(L), "flops": sum(x["flops"] for x in L), "bytes": sum(x["bytes"] for x in L),
                      "avg_flops_per_launch": sum(x["flops"] for x in L) / len(L), "avg_bytes_per_launch": sum(x["bytes"] for x in L) / len(L)}))
