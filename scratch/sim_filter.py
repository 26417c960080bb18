"""CPU model of the tile kernel's filter + parked-key insertion for a uniform cloud: counts lock-step pair
steps and drain rounds per 64-query wave under different drain policies (a planning aid, not product code)."""
import sys
import numpy as np

rng = np.random.default_rng(0)
K = 16
KS = K + 2
occ = 1.16 * 0.2387 * (K + 2.6 * np.sqrt(K) + 1)  # points per cell
G = 24  # cells per axis in the model box
n = int(occ * G ** 3)
pts = rng.random((n, 3)) * G
cell = np.floor(pts).astype(int)
half = (pts[:, 0] - cell[:, 0] >= 0.5).astype(int)
key = ((cell[:, 2] * G + cell[:, 1]) * G + cell[:, 0]) * 2 + half
order = np.argsort(key, kind="stable")
pts, cell, half, key = pts[order], cell[order], half[order], key[order]
cstart = np.searchsorted(key, np.arange(G ** 3 + 1) * 2)       # first record of every cell
cmid = np.searchsorted(key, np.arange(G ** 3) * 2 + 1)          # first high-half record
SEG_DY = [0, -1, 1, 0, 0, -1, 1, -1, 1]
SEG_DZ = [0, 0, 0, -1, 1, -1, -1, 1, 1]
TILE = (5, 3, 2)


def wave_queries():
    """64-lane batches of queries in the kernel's order: tile by tile, rows of the tile, 4 waves share a tile."""
    out = []
    for tz in range(2, G - 2 - TILE[2], TILE[2]):
        for ty in range(2, G - 2 - TILE[1], TILE[1]):
            for tx in range(2, G - 2 - TILE[0], TILE[0]):
                q = []
                for z in range(tz, tz + TILE[2]):
                    for y in range(ty, ty + TILE[1]):
                        c0 = (z * G + y) * G + tx
                        q.extend(range(cstart[c0], cstart[c0 + TILE[0]]))
                q = np.array(q)
                for w in range(0, len(q), 64):
                    out.append(q[w:w + 64])
    return out


def segments(qi):
    """(start, len) of the nine row segments of query qi (x-half trimming on)."""
    cx, cy, cz = cell[qi]
    segs = []
    for dy, dz in zip(SEG_DY, SEG_DZ):
        row = ((cz + dz) * G + (cy + dy)) * G
        if half[qi] == 0:
            s, e = cstart[row + cx - 1], cmid[row + cx + 1]
        else:
            s, e = cmid[row + cx - 1], cstart[row + cx + 2]
        segs.append((s, e - s))
    return segs


def run(policy, slots=16, partial=8, nwaves=150, seg_drain=False, sort_segs=0, seed_reach=0.0):
    waves = wave_queries()
    sel = rng.choice(len(waves), size=min(nwaves, len(waves)), replace=False)
    tot_steps = tot_rounds = tot_drains = tot_parked = tot_cand = 0
    for wi in sel:
        qs = waves[wi]
        L = [[] for _ in qs]          # sorted lists (as python lists of d2)
        thr = [np.inf] * len(qs)
        if seed_reach > 0:
            for li, q in enumerate(qs):
                x, y, z = pts[q] - cell[q]
                rx = (x + 1.0, 1.5 - x) if half[q] == 0 else (x - 0.5 + 1.0, 2.0 - x)
                reach = min(rx[0], rx[1], y + 1.0, 2.0 - y, z + 1.0, 2.0 - z) * seed_reach
                thr[li] = reach * reach
        park = [[] for _ in qs]
        segs = [segments(q) for q in qs]
        if sort_segs == 1:   # own row first, the other eight by length, longest first
            segs = [sg[:1] + sorted(sg[1:], key=lambda a: -a[1]) for sg in segs]
        elif sort_segs == 2:  # own row, then faces sorted, then diagonals sorted
            segs = [sg[:1] + sorted(sg[1:5], key=lambda a: -a[1]) + sorted(sg[5:], key=lambda a: -a[1]) for sg in segs]
        elif sort_segs == 3:  # all nine
            segs = [sorted(sg, key=lambda a: -a[1]) for sg in segs]

        def drain(rounds=None):
            nonlocal tot_rounds, tot_drains
            mx = max(len(p) for p in park)
            r = mx if rounds is None else min(mx, rounds)
            r = max(r, 1)
            tot_rounds += r
            tot_drains += 1
            for li in range(len(qs)):
                take = park[li][-r:] if r < len(park[li]) else park[li]
                park[li] = park[li][:-r] if r < len(park[li]) else []
                if take:
                    L[li] = sorted(L[li] + take)[:KS]
                    if len(L[li]) == KS:
                        thr[li] = min(thr[li], L[li][-1])

        for seg in range(9):
            tmax = max(s[seg][1] for s in segs)
            for t in range(0, tmax, 2):
                tot_steps += 1
                for li, q in enumerate(qs):
                    s, ln = segs[li][seg]
                    for j in (t, t + 1):
                        if j < ln:
                            d = ((pts[q] - pts[s + j]) ** 2).sum()
                            tot_cand += 1
                            if d < thr[li]:
                                park[li].append(d)
                                tot_parked += 1
                if max(len(p) for p in park) >= slots - 1:
                    if policy == "full":
                        drain()
                    elif policy == "quantile":
                        hs = sorted(len(p) for p in park)
                        drain(max(2, hs[int(len(hs) * partial / 100.0)]))
                    else:
                        drain(partial)
            if seg_drain and seg < 8:
                drain() if policy == "full" else drain(partial)
        drain()  # final: everything
        while any(park):
            drain()
    nw = len(sel)
    return dict(steps=tot_steps / nw, rounds=tot_rounds / nw, drains=tot_drains / nw,
                parked=tot_parked / nw / 64, cand=tot_cand / nw / 64)


if __name__ == "__main__":
    for args in [dict(policy="partial", partial=8, seed_reach=1.0, sort_segs=2), dict(policy="quantile", partial=50, seed_reach=1.0, sort_segs=2),
                 dict(policy="quantile", partial=75, seed_reach=1.0, sort_segs=2), dict(policy="quantile", partial=30, seed_reach=1.0, sort_segs=2),
                 dict(policy="partial", partial=5, seed_reach=1.0, sort_segs=2), dict(policy="partial", partial=4, seed_reach=1.0, sort_segs=2)]:
        rng = np.random.default_rng(1)
        print(args, run(**args))
