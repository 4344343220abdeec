#!/usr/bin/env python3
"""Would batching phase B's lobe code through a second LDS queue pay?  

Model of the split kernel's pass loop with the rates of the lane census (profiles/r06_experiments/split_census_r06a_*.log): 95 % of a
batch's 64 primaries continue, a secondary segment continues with probability 0.335, depth limit 5.  Variant: after a phase-B scan every
lane is free again -- a path that ends is added to the pixel, a path that continues leaves its HIT in a queue of `cap` entries --, free lanes
are filled first with queued rays, then with queued hits (which run the lobe code, phase C, and go on as rays), and phase A runs when the
queues leave room for its 64 rays.  Costs in VALU instructions per execution from the listing: A 360 (with its own lobe code), a phase-B pass
330, a lobe execution 160; today's loop: 360 + 1.452 * 330 + 1.26 * 160 = 1041 per 64 samples.

    cap   take hits when >=    B passes / 64 samples (lanes)   lobe executions (lanes)   cost
    80..  16                   2.03 (0.70)                     1.03 (0.46)              1196
    96    32                   1.73 (0.82)                     0.73 (0.64)              1049
    96    48                   1.72 (0.83)                     0.72 (0.66)              1042
    112   48                   1.53 (0.93)                     0.53 (0.90)               947   <- pays: -9 %
    128   48                   1.53 (0.93)                     0.53 (0.90)               947

A hit entry is 88 B; five waves per SIMD leave a wave 7.6 KB of LDS: 86 entries.  Not built.
"""
import random


def sim(QCAP, TH, batches=3000, pA=0.95, pB=0.335, maxdepth=5, seed=1):
    random.seed(seed)
    rq, hq = [], []
    nA = nB = nC = 0
    lanesB = lanesC = 0
    remaining = batches
    while True:
        live = []
        if not rq and remaining > 0 and len(hq) + 64 <= QCAP and not (len(hq) >= TH):
            remaining -= 1
            nA += 1
            rq += [2] * sum(random.random() < pA for _ in range(64))
        nr = min(64, len(rq))
        live += rq[len(rq) - nr:]
        del rq[len(rq) - nr:]
        free = 64 - nr
        if free > 0 and hq:
            can_A_later = remaining > 0 and len(hq) + 64 <= QCAP
            if len(hq) >= min(free, TH) or not can_A_later or (remaining == 0 and not rq):
                nh = min(free, len(hq))
                if nh >= (TH if (can_A_later and nr > 0) else 1) or not can_A_later:
                    take = hq[len(hq) - nh:]
                    del hq[len(hq) - nh:]
                    nC += 1
                    lanesC += nh
                    live += [d + 1 for d in take]
        if not live:
            if not rq and not hq and remaining == 0:
                break
            continue
        nB += 1
        lanesB += len(live)
        hq += [d for d in live if d <= maxdepth and random.random() < pB]
    return nA, nB, nC, lanesB / nB / 64, lanesC / max(nC, 1) / 64


if __name__ == "__main__":
    for QCAP in (80, 96, 112, 128):
        for TH in (16, 32, 48):
            nA, nB, nC, oB, oC = sim(QCAP, TH)
            print(f"cap {QCAP:3d} threshold {TH:2d}: B passes {nB / nA:.3f} at {oB:.3f} of lanes, lobe executions {nC / nA:.3f} at {oC:.3f}, "
                  f"cost {(nA * 360 + nB * 330 + nC * 160) / nA:.0f} (today 1041)")
