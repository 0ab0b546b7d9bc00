"""Estimate of what og_group_voices buys on the synthetic note plans (a MODEL of instruction counts, not a measurement).

For groups of 64 consecutive voice slots (= one workgroup of the four-wave fm kernel) and every 8-frame chunk: which body each
envelope wave runs -- release-free, release, checked -- from the voices' note events and the envelopes' stage lengths; cost of
a body = its static VALU count (scripts/isa_blocks.py).  Reported: the chunk shares, the mean cost per workgroup, and the
busiest CU when workgroup i goes to CU i mod 256 (all workgroups of a 65 536-voice launch are resident at once, so a launch
ends with the busiest CU) -- for voices in voice order, og_group_voices policy 1 (by first note-off) and policy 2 (policy 1's
groups dealt out by event count).

    python scripts/group_model.py [voices]        (default 65536; takes a minute)
"""
import sys

import numpy as np

sys.path.insert(0, __import__("os").path.dirname(__import__("os").path.dirname(__import__("os").path.abspath(__file__))))
import oscen_amd  # noqa: E402

V = int(sys.argv[1]) if len(sys.argv) > 1 else 65536
ENVS = {"w0": [(480, 4800, 14400)], "w1": [(480, 4800, 14400)], "w2": [(480, 9600, 24000), (480, 9600, 14400)]}  # (attack, decay, release) frames
BASE = {"w0": 201.0, "w1": 187.0, "w2": 194.0}   # release-free chunk, VALU per 8 frames (body + loop tail)
REL = {"w0": 245.0, "w1": 245.0, "w2": 232.0}    # release body
CHK = 2.0                                          # checked body = this x the release-free one
W3 = 8 * 15.0                                      # the filter wave


def chunk_classes(plans, T):
    ev_v, ev_f, ev_x = plans["events"]
    keep = ev_f < T
    ev_v, ev_f, ev_x = ev_v[keep], ev_f[keep], ev_x[keep]
    nch = T // 8
    idx = np.argsort(ev_v, kind="stable")
    ev_v, ev_f, ev_x = ev_v[idx], ev_f[idx], ev_x[idx]
    starts = np.searchsorted(ev_v, np.arange(V))
    ends = np.searchsorted(ev_v, np.arange(V), side="right")
    out = {}
    for w, es in ENVS.items():
        rel = np.zeros((V, nch), dtype=bool)
        chk = np.zeros((V, nch), dtype=bool)
        for v in range(V):
            fs, xs = ev_f[starts[v]:ends[v]], ev_x[starts[v]:ends[v]]
            for (A, D, R) in es:
                for k in range(len(fs)):
                    f = int(fs[k])
                    nxt = int(fs[k + 1]) if k + 1 < len(fs) else T
                    chk[v, min(f // 8, nch - 1)] = True
                    if xs[k] > 0:
                        for b in (f + A, f + A + D):
                            if b < nxt and b < T:
                                chk[v, b // 8] = True
                    else:
                        e = min(f + R, nxt, T)
                        rel[v, f // 8:(e + 7) // 8] = True
                        if f + R < nxt and f + R < T:
                            chk[v, (f + R) // 8] = True
        out[w] = (rel, chk)
    n_ev = np.bincount(ev_v, minlength=V)
    first_off = np.full(V, 1 << 40, dtype=np.int64)
    off = ev_x <= 0
    np.minimum.at(first_off, ev_v[off], ev_f[off])
    first_any = np.full(V, 1 << 40, dtype=np.int64)
    np.minimum.at(first_any, ev_v, ev_f)
    return out, n_ev, first_off, first_any


def evaluate(classes, slot_order):
    """slot_order[r] = voice in slot r"""
    Wg = V // 64
    nch = next(iter(classes.values()))[0].shape[1]
    wg_cost = np.full(Wg, W3 * nch)
    shares = {}
    for w, (rel, chk) in classes.items():
        relw = rel[slot_order].reshape(Wg, 64, nch).any(1)
        chkw = chk[slot_order].reshape(Wg, 64, nch).any(1)
        r = relw & ~chkw
        q = ~relw & ~chkw
        wg_cost = wg_cost + q.sum(1) * BASE[w] + r.sum(1) * REL[w] + chkw.sum(1) * CHK * BASE[w]
        shares[w] = (round(float(q.mean()), 3), round(float(r.mean()), 3), round(float(chkw.mean()), 3))
    cu = np.zeros(256)
    np.add.at(cu, np.arange(Wg) % 256, wg_cost)
    return shares, float(wg_cost.mean()) * Wg / 256, float(cu.max())


def orders(n_ev, first_off, first_any):
    ident = np.arange(V)
    p1 = np.lexsort((first_any, first_off))
    Wg = V // 64
    weight = n_ev[p1].reshape(Wg, 64).sum(1)
    rank = np.argsort(-weight, kind="stable")
    place = np.empty(Wg, dtype=np.int64)
    for k in range(Wg):
        row, col = divmod(k, 256)
        place[k] = row * 256 + ((255 - col) if (row & 1) else col)
    pos = np.empty(Wg, dtype=np.int64)
    pos[rank] = place
    p2 = np.empty(V, dtype=np.int64)
    p2[(pos[:, None] * 64 + np.arange(64)[None, :]).reshape(-1)] = p1
    return {"voice order": ident, "policy 1": p1, "policy 2": p2}


for name, span, fold in (("driver's command (20 blocks, slice of the score)", 5120, "slice"), ("1 s default run", 48128, "scale")):
    plans = oscen_amd.note_plans(V, span=span if span < 48000 else 0, fold=fold)
    classes, n_ev, first_off, first_any = chunk_classes(plans, span)
    print(name)
    ref = None
    for oname, order in orders(n_ev, first_off, first_any).items():
        shares, mean_cu, max_cu = evaluate(classes, order)
        ref = ref or max_cu
        print("  %-12s (release-free, release, checked) shares %s  mean CU %.3g  busiest CU %.3g  -> x%.3f against voice order" % (
            oname, shares["w0"], mean_cu, max_cu, ref / max_cu))
