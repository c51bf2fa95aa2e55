"""NumPy restatement of the tiling rule of `gconv-adapter_amd/csrc/tiles.h` (test infrastructure:
the GPU tests compare every array of a built plan with it, integer for integer)."""
from __future__ import annotations

import numpy as np

HALO_NODES, HALO_EDGES, HALO_SLOT = 64, 256, 456
HALO_RP, HALO_COL, HALO_DIS = 64, 132, 388


def _reach(rowptr, col, n):
    mx = np.full(n, -1, dtype=np.int64)
    mn = np.full(n, np.iinfo(np.int32).max, dtype=np.int64)
    nz = np.flatnonzero(rowptr[1:] > rowptr[:-1])
    mn[nz] = col[rowptr[nz]]
    mx[nz] = col[rowptr[nz + 1] - 1]
    return mx, mn


def cuts(rowptr, col, n, T, slack):
    """tile_ptr: cut k is the candidate closest to k * T in (k * T - slack, k * T] that no edge of the
    rows within reach (128 below, 64 above k * T) crosses; k * T if there is none."""
    ntiles = -(-max(n, 1) // T)
    mx, mn = _reach(rowptr, col, n)
    out = [0]
    for k in range(1, ntiles):
        q0 = k * T
        pick = q0
        for q in range(q0, max(q0 - slack, q0 - 64), -1):
            if q < 1:
                break
            below = mx[max(q0 - 128, 0):q]
            above = mn[q:min(q0 + 64, n)]
            if not ((below >= q).any() or (above < q).any()):
                pick = q
                break
        out.append(pick)
    out.append(n)
    return np.asarray(out, dtype=np.int64)


def tile(rowptr, col, ts, te, dis=None):
    """(record fields, local columns of the own rows, halo nodes, halo rowptr, halo local columns,
    overflow) of the tile [ts, te) — halo nodes in order of first appearance, halo1 first."""
    nt = te - ts
    halo, overflow = [], False

    def add(c):
        nonlocal overflow
        if c not in halo:
            if len(halo) < HALO_NODES:
                halo.append(c)
            else:
                overflow = True

    e0, e1 = int(rowptr[ts]), int(rowptr[te])
    for c in col[e0:e1]:
        if c < ts or c >= te:
            add(int(c))
    nh1 = len(halo)
    rows = [(int(rowptr[h]), int(rowptr[h + 1])) for h in halo[:nh1]]
    he = sum(b - a for a, b in rows)
    if he > HALO_EDGES or e1 - e0 > (1 << 14):
        overflow = True
    else:    # halo2 in (edge position, halo row) order: the halo1 rows are walked side by side
        for k in range(max([b - a for a, b in rows], default=0)):
            for a, b in rows:
                if a + k < b:
                    c = int(col[a + k])
                    if c < ts or c >= te:
                        add(c)

    def local(c):
        return c - ts if ts <= c < te else (nt + halo.index(c) if c in halo else -1)
    lcol = np.asarray([local(int(c)) for c in col[e0:e1]], dtype=np.int64)
    hrp, hcol = [0], []
    for h in range(nh1 if not overflow else 0):
        a, b = int(rowptr[halo[h]]), int(rowptr[halo[h] + 1])
        hcol += [local(int(c)) for c in col[a:b]]
        hrp.append(len(hcol))
    rec = dict(first=ts, count=nt, n1=nt + nh1, n0=nt + len(halo), e0=e0, ne_own=e1 - e0, ne_halo=he)
    hdis = None if dis is None else np.asarray([dis[h] for h in halo], dtype=np.float32)
    return rec, lcol, np.asarray(halo, dtype=np.int64), np.asarray(hrp), np.asarray(hcol[:HALO_EDGES]), hdis, overflow


def check_plan(plan, rowptr, col, n, T, slack, transposed=False, dis=None, order_by=None):
    """Asserts that `plan`'s tile arrays of one direction equal the restatement; returns the list of
    per-tile records (dicts) in tile order. `order_by`: the forward records (both directions are
    listed in the order of the forward tiles' sizes)."""
    tp = cuts(np.asarray(plan.fwd.rowptr.cpu()), np.asarray(plan.fwd.col.cpu())[:plan.fwd.nnz], n, T, slack)
    got_tp = plan.tile_ptr.cpu().numpy()
    np.testing.assert_array_equal(got_tp, tp)
    recs = plan.tiles(transposed).cpu().numpy()
    lcol = plan.tile_lcol(transposed).cpu().numpy()
    halo = plan.tile_halo(transposed).cpu().numpy()
    want, refs, any_overflow = [], [], False
    for t in range(len(tp) - 1):
        rec, wl, wh, whrp, whcol, whdis, ov = tile(rowptr, col, int(tp[t]), int(tp[t + 1]), dis)
        any_overflow |= ov
        rec["slot"] = t
        want.append(rec)
        refs.append((wl, wh, whrp, whcol, whdis))
    assert (plan.n_tiles == 0) == any_overflow or n == 0
    if any_overflow:      # (one tile given up: the builder stops, nothing else is defined)
        return want
    for t, (rec, (wl, wh, whrp, whcol, whdis)) in enumerate(zip(want, refs)):
        np.testing.assert_array_equal(lcol[rec["e0"]:rec["e0"] + rec["ne_own"]], wl, err_msg=f"lcol of tile {t}")
        slot = halo[t]
        np.testing.assert_array_equal(slot[:len(wh)], wh, err_msg=f"halo nodes of tile {t}")
        if rec["n1"] > rec["count"]:
            np.testing.assert_array_equal(slot[HALO_RP:HALO_RP + len(whrp)], whrp, err_msg=f"halo rowptr of tile {t}")
            np.testing.assert_array_equal(slot[HALO_COL:HALO_COL + len(whcol)], whcol, err_msg=f"halo cols of tile {t}")
        if whdis is not None and len(wh):
            np.testing.assert_array_equal(slot[HALO_DIS:HALO_DIS + len(wh)].view(np.float32), whdis)
    key = order_by if order_by is not None else want
    if len(want) <= 256:       # one tile per persistent workgroup: node order, not sorted
        order = list(range(len(want)))
    else:
        order = sorted(range(len(want)), key=lambda t: (-key[t]["n0"], t))  # largest first, stable
    fields = ("first", "count", "n1", "n0", "slot", "e0", "ne_own", "ne_halo")
    np.testing.assert_array_equal(recs, np.asarray([[want[t][f] for f in fields] for t in order]))
    return want
