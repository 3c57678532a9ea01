"""Prototype + checker of the NTT tile planner (the C++ planner in rust-kzg_amd/csrc/ntt_plan.h is a port of
plan_pass() below; tests/test_ntt_plan_cpu.py compares the two table for table).

A pass of the transform runs T butterfly stages of the DIT network on tiles of 4096 elements.  Tile-local element
index idx (12 bits): bits [0, T) are the stage bits (stage s pairs idx and idx ^ (1 << s)), bits [T, 12) are column
bits (independent sub-problems).  1024 threads (16 waves x 64 lanes) hold 4 elements each per round; a round runs two
stages (M = 2: elements idxA, idxA | 1 << pos, idxB = idxA | 2 << pos, idxB | 1 << pos) or one (M = 1: two unrelated
pairs).  A phase = consecutive rounds in which every wave keeps the same 256 elements: its exchanges go through LDS
without a workgroup barrier.  The planner picks, per round, which idx bit every thread-id bit stands for:
  * wave bits (4) are fixed per phase,
  * lane bits (6): in the round that loads from / stores to global memory the low lane bits follow the low address
    bits (coalescing), elsewhere they are chosen so that the 32 lanes of a ds_read_b32 group hit 32 distinct banks
    under the fixed XOR swizzle swz().
Run: python tools/ntt_plan_sim.py   (simulates every plan against a direct transform over a small prime field)
"""
import itertools
import sys

LOGT = 12
TILE = 1 << LOGT
KIND_A1, KIND_A2, KIND_B = 0, 1, 2


SWZ_COL = [23, 29, 19, 10, 19, 13, 27]  # found by search_swizzle() below


def swz(i):
    """LDS position of tile element i: XOR-linear, bits 5..11 unchanged (bijection)."""
    x = 0
    for k in range(7):
        if (i >> (5 + k)) & 1:
            x ^= SWZ_COL[k]
    return i ^ x


def bank(i):
    return swz(i) & 31


def addr_rank(kind, T, bit, store):
    """significance of idx bit `bit` in the global address of the load (store=False) / store round; lower = lower address bit"""
    stage = bit < T
    if kind == KIND_A1:
        if store:
            return bit
        return (T - 1 - bit) if stage else bit
    if kind == KIND_A2:
        if store:
            return bit if stage else 100 + bit
        return (bit - T) if not stage else 50 + (T - 1 - bit)
    # KIND_B: address = origin + (r << s0) + c
    return (bit - T) if not stage else 50 + bit


def phases_for(kind, T):
    """-> list of (stage_lo, stage_hi, free_bits[8], wave_bits[4])"""
    allbits = list(range(LOGT))

    def mk(lo, hi, prefer):
        # the 8 bits a wave keeps to itself in this phase: the stage bits, then bits from `prefer` (in order)
        F = list(range(lo, hi))
        for b in prefer + allbits:
            if len(F) < 8 and b not in F:
                F.append(b)
        assert len(F) == 8 and len(set(F)) == 8, (kind, T, lo, hi, prefer)
        W = [b for b in allbits if b not in F]
        return (lo, hi, F, W)

    if kind == KIND_A1:
        if T <= 8:
            return [mk(0, T, list(range(T, 8)))]
        return [mk(0, 6, [T - 2, T - 1]), mk(6, T, [])]
    assert T <= 10
    if T <= 6:
        return [mk(0, T, list(range(T, 8)))]
    if kind == KIND_A2:
        return [mk(0, 6, [T, T + 1]), mk(6, T, [])]
    return [mk(0, 6, [T, T + 1]), mk(6, T, [T, T + 1])]


def rounds_of_phase(lo, hi):
    r = []
    s = lo
    while s < hi:
        m = 2 if s + 2 <= hi else 1
        r.append((s, m))
        s += m
    return r


def brev_map(T):
    """LDS position of DIT position idx when the tile holds the data in NATURAL order (the forward half of the fused DAS
    extension reads what the inverse half left): the low T bits bit-reversed, column bits unchanged (XOR-linear)"""
    mT = (1 << T) - 1

    def f(i):
        p, r = i & mT, 0
        for k in range(T):
            if (p >> k) & 1:
                r |= 1 << (T - 1 - k)
        return (i & ~mT) | r
    return f


def conflicts(bits5, posmap=None):
    """extra LDS cycles of a 32-lane group whose lanes vary the idx bits `bits5` (others fixed at 0)"""
    seen = {}
    for l in range(32):
        i = 0
        for k, b in enumerate(bits5):
            if (l >> k) & 1:
                i |= 1 << b
        if posmap:
            i = posmap(i)
        seen[bank(i)] = seen.get(bank(i), 0) + 1
    return max(seen.values()) - 1


def plan_pass(kind, T, first_io=True, last_io=True, posmap=None):
    """-> dict(rounds=[dict(pos, M, barrier_after, lane_bits[6], pair_bit or None, wave_bits[4])], tab=[[(idxA, idxB)]*1024])
    first_io / last_io: the first round loads from / the last stores to global memory (lane bits follow the addresses);
    posmap: LDS position of an index (identity when None)"""
    ph = phases_for(kind, T)
    rounds = []
    for pi, (lo, hi, F, W) in enumerate(ph):
        rs = rounds_of_phase(lo, hi)
        for ri, (pos, M) in enumerate(rs):
            rounds.append(dict(pos=pos, M=M, F=F, W=W, last_of_phase=(ri == len(rs) - 1)))
    while len(rounds) < 2:  # the kernel peels a loading and a storing round: T <= 2 gets rounds without stages (copies)
        lo, hi, F, W = ph[-1]
        if rounds:
            rounds[-1]["last_of_phase"] = False
        rounds.append(dict(pos=0, M=0, F=F, W=W, last_of_phase=True))
    n = len(rounds)
    for r, R in enumerate(rounds):
        pos, M, F, W = R["pos"], R["M"], R["F"], R["W"]
        stage_bits = [pos + k for k in range(M)]
        rest = [b for b in F if b not in stage_bits]  # 6 (M=2), 7 (M=1) or 8 (M=0) bits
        first, last = r == 0, r == n - 1
        R["barrier_after"] = (not last) and R["last_of_phase"]
        npair = len(rest) - 6  # bits that tell the thread's elements apart besides the stage bits
        best = None
        if (first and first_io) or (last and last_io):
            rest_sorted = sorted(rest, key=lambda b: addr_rank(kind, T, b, store=not (first and first_io)))
            # the highest-ranked bits tell the elements of a thread apart, the others are the lanes, low address bits first
            lanes, pair = rest_sorted[:6], rest_sorted[6:]
            best = (lanes, pair)
        else:
            for pair in itertools.combinations(rest, npair):
                lanes6 = [b for b in rest if b not in pair]
                for hi_lane in lanes6:
                    lanes5 = [b for b in lanes6 if b != hi_lane]
                    c = conflicts(lanes5, posmap)
                    cand = (c, lanes5 + [hi_lane], list(pair))
                    if best is None or cand[0] < best[0]:
                        best = cand
            best = (best[1], best[2])
        R["lane_bits"], R["pair_bits"] = best
        R["conflicts"] = conflicts(R["lane_bits"][:5], posmap)
    tab = []
    for R in rounds:
        t = []
        for u in range(1024):
            l, w = u & 63, u >> 6
            base = 0
            for k, b in enumerate(R["lane_bits"]):
                if (l >> k) & 1:
                    base |= 1 << b
            for k, b in enumerate(R["W"]):
                if (w >> k) & 1:
                    base |= 1 << b
            M, pos, pb = R["M"], R["pos"], R["pair_bits"]
            if M == 2:
                e = [base, base | 1 << pos, base | 2 << pos, base | 3 << pos]
            elif M == 1:
                e = [base, base | 1 << pos, base | 1 << pb[0], base | 1 << pb[0] | 1 << pos]
            else:
                e = [base, base | 1 << pb[0], base | 1 << pb[1], base | 1 << pb[0] | 1 << pb[1]]
            t.append(tuple(e))
        tab.append(t)
    return dict(kind=kind, T=T, rounds=rounds, tab=tab)


def plan_das(T):
    """The DAS extension of lists of 2^T <= 4096 elements in ONE tile pass: the rounds of the inverse transform, whose
    last round multiplies result j by the twist and leaves the tile in LDS in natural order, then the rounds of the
    forward transform reading it at bit-reversed positions.  Every round carries: part (0 inverse, 1 forward),
    twist (multiply before the LDS store), unit (stages 0 / 1 at position 0 need no multiplication), and the forward
    rounds the position map."""
    inv = plan_pass(KIND_A1, T, first_io=True, last_io=False)
    fwd = plan_pass(KIND_A1, T, first_io=False, last_io=True, posmap=brev_map(T))
    rounds, tab = [], []
    for r, R in enumerate(inv["rounds"]):
        R = dict(R, part=0, unit=(r == 0), twist=(r == len(inv["rounds"]) - 1))
        if R["twist"]:
            R["barrier_after"] = True
        rounds.append(R)
    for r, R in enumerate(fwd["rounds"]):
        rounds.append(dict(R, part=1, unit=(r == 0), twist=False))
    return dict(kind=KIND_A1, T=T, rounds=rounds, tab=inv["tab"] + fwd["tab"], das=True)


# ------------------------------------------------------------------------------------------ simulation
P = 2013265921  # 15 * 2^27 + 1
G = 31


def root_of_unity(order):
    assert (P - 1) % order == 0
    return pow(G, (P - 1) // order, P)


def brev(v, bits):
    r = 0
    for k in range(bits):
        if (v >> k) & 1:
            r |= 1 << (bits - 1 - k)
    return r


def run_tile(plan, load, store, tw, twist=None):
    """load(idx) -> value, store(idx, value), tw(stage s, idx of the lower element[, part]) -> twiddle,
    twist(idx) -> multiplier (fused DAS plans)"""
    lds = {}
    owner = {}
    nr = len(plan["rounds"])
    T = plan["T"]
    bmap = brev_map(T)
    for r, R in enumerate(plan["rounds"]):
        newlds, newowner = {}, {}
        part = R.get("part", 0)
        pm = bmap if part == 1 else (lambda i: i)
        for u in range(1024):
            idx = plan["tab"][r][u]
            wave = u >> 6
            if r == 0:
                e = [load(i) for i in idx]
            else:
                e = [lds[pm(i)] for i in idx]
                if not plan["rounds"][r - 1]["barrier_after"]:
                    for i in idx:
                        assert owner[pm(i)] == wave, "wave-local exchange reads another wave's element"
            M, pos = R["M"], R["pos"]
            twf = (lambda s_, i_: tw(s_, i_, part)) if plan.get("das") else tw
            if M >= 1:
                for a, b in ((0, 1), (2, 3)):
                    assert idx[b] == idx[a] | (1 << pos) and not (idx[a] >> pos) & 1
                    t = e[b] * twf(pos, idx[a]) % P
                    e[a], e[b] = (e[a] + t) % P, (e[a] - t) % P
            if M == 2:
                for a, b in ((0, 2), (1, 3)):
                    assert idx[b] == idx[a] | (2 << pos) and not (idx[a] >> (pos + 1)) & 1
                    t = e[b] * twf(pos + 1, idx[a]) % P
                    e[a], e[b] = (e[a] + t) % P, (e[a] - t) % P
            if R.get("twist"):
                e = [v * twist(i) % P for v, i in zip(e, idx)]
            for i, v in zip(idx, e):
                if r == nr - 1:
                    store(i, v)
                else:
                    assert pm(i) not in newlds
                    newlds[pm(i)] = v
                    newowner[pm(i)] = wave
        lds, owner = newlds, newowner


def das_sim(x, L):
    """DAS extension of x (len 2^L <= 2048) through the fused plan; = fft(w2n^j * ifft(x)_j)"""
    n = 1 << L
    w = root_of_unity(n) if n > 1 else 1
    w2 = root_of_unity(2 * n)
    plan = plan_das(L)
    data = list(x) + [0] * (TILE - n)
    out = [None] * TILE
    ninv = pow(n, P - 2, P)

    def load(i):
        c, p = i >> L, i & (n - 1)
        return data[(c << L) + brev(p, L)]

    def store(i, v):
        out[i] = v * ninv % P

    def tw(s, i, part):
        j = i & ((1 << s) - 1)
        e = j * (n >> (s + 1))
        return pow(w, (n - e) % n if part == 0 else e, P)

    def twist(i):
        return pow(w2, i & (n - 1), P)

    run_tile(plan, load, store, tw, twist)
    return out[:n]


def das_ref(x, L):
    n = 1 << L
    ninv = pow(n, P - 2, P)
    w = root_of_unity(n) if n > 1 else 1
    winv = pow(w, P - 2, P) if n > 1 else 1
    w2 = root_of_unity(2 * n)
    # inverse transform = forward with w^-1, scaled
    a = [x[brev(i, L)] for i in range(n)]
    for s_ in range(L):
        half = 1 << s_
        ws = pow(winv, n >> (s_ + 1), P)
        for blk in range(0, n, 2 * half):
            t = 1
            for j in range(half):
                u, v = a[blk + j], a[blk + j + half] * t % P
                a[blk + j], a[blk + j + half] = (u + v) % P, (u - v) % P
                t = t * ws % P
    a = [v * ninv % P * pow(w2, j, P) % P for j, v in enumerate(a)]
    return ntt_ref(a, L)


def split_passes(L):
    """stage counts per pass for n = 2^L > 4096 (every pass at most 10 stages: >= 4 columns per tile)"""
    np_ = (L + 9) // 10
    base, extra = divmod(L, np_)
    return [base + (1 if i < extra else 0) for i in range(np_)]


def ntt_sim(x, L):
    """forward transform of x (len 2^L) through the planned passes; returns natural-order output"""
    n = 1 << L
    w = root_of_unity(n) if n > 1 else 1
    if L <= LOGT:
        plan = plan_pass(KIND_A1, L)
        total = n
        # pad the batch to whole tiles (C transforms per tile)
        C = TILE >> L
        data = list(x) + [0] * (TILE - n) if n < TILE else list(x)
        out = [None] * len(data)

        def load(i):
            c, p = i >> L, i & (n - 1)
            return data[(c << L) + brev(p, L)]

        def store(i, v):
            out[i] = v

        def tw(s, i):
            j = i & ((1 << s) - 1)
            return pow(w, j * (n >> (s + 1)), P)

        run_tile(plan, load, store, tw)
        return out[:n]
    Ts = split_passes(L)
    TA = Ts[0]
    Lh = L - TA
    C = TILE >> TA
    buf = [None] * n
    planA = plan_pass(KIND_A2, TA)
    for tile in range(n >> LOGT):
        o_base = tile * C

        def load(i):
            c, p = i >> TA, i & ((1 << TA) - 1)
            return x[(o_base + c) + (brev(p, TA) << Lh)]

        def store(i, v):
            c, p = i >> TA, i & ((1 << TA) - 1)
            buf[(brev(o_base + c, Lh) << TA) + p] = v

        def tw(s, i):
            j = i & ((1 << s) - 1)
            return pow(w, j * (n >> (s + 1)), P)

        run_tile(planA, load, store, tw)
    s0 = TA
    for TB in Ts[1:]:
        planB = plan_pass(KIND_B, TB)
        Cb = TILE >> TB
        nxt = [None] * n
        lo_tiles = (1 << s0) // Cb
        for tile in range(n >> LOGT):
            hi, lo0 = tile // lo_tiles, (tile % lo_tiles) * Cb
            origin = (hi << (s0 + TB)) + lo0

            def load(i):
                c, r = i >> TB, i & ((1 << TB) - 1)
                return buf[origin + (r << s0) + c]

            def store(i, v):
                c, r = i >> TB, i & ((1 << TB) - 1)
                nxt[origin + (r << s0) + c] = v

            def tw(s, i):
                c, r = i >> TB, i & ((1 << TB) - 1)
                jg = ((r & ((1 << s) - 1)) << s0) + lo0 + c
                return pow(w, jg * (n >> (s0 + s + 1)), P)

            run_tile(planB, load, store, tw)
        buf = nxt
        s0 += TB
    return buf


def ntt_ref(x, L):
    n = 1 << L
    if n == 1:
        return list(x)
    w = root_of_unity(n)
    a = [x[brev(i, L)] for i in range(n)]
    for s in range(L):
        half = 1 << s
        ws = pow(w, n >> (s + 1), P)
        for blk in range(0, n, 2 * half):
            t = 1
            for j in range(half):
                u, v = a[blk + j], a[blk + j + half] * t % P
                a[blk + j], a[blk + j + half] = (u + v) % P, (u - v) % P
                t = t * ws % P
    return a


def search_swizzle(seed=1, trials=6):
    """hill-climb over the seven 5-bit XOR constants of swz(): cost = bank conflicts summed over every round of every plan"""
    global SWZ_COL
    import random

    def cost(cols):
        global SWZ_COL
        SWZ_COL = cols
        tot = worst = 0
        for kind in (KIND_A1, KIND_A2, KIND_B):
            for T in range(0, 13 if kind == KIND_A1 else 11):
                for R in plan_pass(kind, T)["rounds"]:
                    tot += R["conflicts"] * (4 if T >= 7 else 1)
                    worst = max(worst, R["conflicts"])
        return tot, worst

    rnd = random.Random(seed)
    best = None
    for _ in range(trials):
        cols = [rnd.randrange(32) for _ in range(7)]
        c = cost(cols)
        improved = True
        while improved:
            improved = False
            for k in range(7):
                for v in range(32):
                    if v != cols[k]:
                        n = cols[:]
                        n[k] = v
                        cn = cost(n)
                        if cn < c:
                            c, cols, improved = cn, n, True
        if best is None or c < best[0]:
            best = (c, cols)
            print(best, flush=True)
    SWZ_COL = best[1]
    return best


def describe(plan):
    out = []
    for R in plan["rounds"]:
        out.append("pos=%d M=%d lanes=%s pair=%s wave=%s barrier=%d conflicts=%d" % (
            R["pos"], R["M"], R["lane_bits"], R["pair_bits"], R["W"], R["barrier_after"], R["conflicts"]))
    return "\n   ".join(out)


if __name__ == "__main__":
    import random

    rnd = random.Random(5)
    verbose = "-v" in sys.argv
    worst = 0
    for kind in (KIND_A1, KIND_A2, KIND_B):
        for T in range(0, 13 if kind == KIND_A1 else 11):
            pl = plan_pass(kind, T)
            c = max(R["conflicts"] for R in pl["rounds"])
            worst = max(worst, c)
            if verbose or c:
                print("kind %d T %2d\n   %s" % (kind, T, describe(pl)))
    print("worst bank conflict over all plans:", worst)
    for L in list(range(0, 13)) + [13, 14, 15, 16]:
        x = [rnd.randrange(P) for _ in range(1 << L)]
        got = ntt_sim(x, L)
        assert got == ntt_ref(x, L), L
        print("L = %2d ok  (passes %s)" % (L, [L] if L <= LOGT else split_passes(L)))
    dworst = 0
    for L in range(0, 13):
        pl = plan_das(L)
        dworst = max(dworst, max(R["conflicts"] for R in pl["rounds"]))
        x = [rnd.randrange(P) for _ in range(1 << L)]
        assert das_sim(x, L) == das_ref(x, L), L
        if verbose:
            print("das T %2d\n   %s" % (L, describe(pl)))
    print("fused DAS plans ok for 2^0 .. 2^12, worst bank conflict", dworst)
