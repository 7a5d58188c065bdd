"""Position-parallel restatement of mm_sketch's emission rule (the decision logic of lrge_amd/csrc/k_sketch_tile.h).

mm_sketch (mm2:sketch.c; oracle/lrge_oracle.c sketch_into) walks a read with a ring of the last w (x, y) infos and a running
minimum.  Its state before step t is a pure function of the infos of steps t-w .. t-1 (the running minimum is the RIGHT-MOST
SMALLEST of them), so whether step p's info is ever written out can be decided from x[p-w .. p+w] and the valid-step counts
l[p+1 .. p+w] alone -- no sequential replay.  The tile kernel evaluates exactly these predicates, one lane per step; this
module states them in plain Python so that they can be checked against the oracle's state machine on the CPU
(tests/test_sketch_model.py) before the kernel is trusted with them.

Steps: without HPC a step is a base; with HPC a step is a homopolymer run (or one ambiguous base).
"""
MAXX = (1 << 64) - 1
MASK64 = (1 << 64) - 1


def hash64(key, mask):
    key = (~key + (key << 21)) & mask
    key = key ^ key >> 24
    key = ((key + (key << 3)) + (key << 8)) & mask
    key = key ^ key >> 14
    key = ((key + (key << 2)) + (key << 4)) & mask
    key = key ^ key >> 28
    key = (key + (key << 31)) & mask
    return key


_NT4 = {c: i for i, c in enumerate("ACGT")}
_NT4.update({c.lower(): i for c, i in list(_NT4.items())})
_NT4["U"] = 3
_NT4["u"] = 3


def steps_of(seq, hpc):
    """[(code, first base, last base)] -- code 4 = ambiguous"""
    out = []
    i, n = 0, len(seq)
    while i < n:
        c = _NT4.get(seq[i], 4)
        j = i
        if hpc and c < 4:
            while j + 1 < n and _NT4.get(seq[j + 1], 4) == c:
                j += 1
        out.append((c, i, j))
        i = j + 1
    return out


def infos_of(seq, w, k, hpc):
    """per step: x (MAXX = none), y low word (pos << 1 | strand), l (valid steps up to and including this one)"""
    st = steps_of(seq, hpc)
    mask = (1 << (2 * k)) - 1
    xs, ys, ls = [], [], []
    last_n = -1
    for t, (c, b0, b1) in enumerate(st):
        if c == 4:
            last_n = t
        l = t - last_n
        x, y = MAXX, 0
        if c < 4 and l >= k:
            first = st[t - k + 1][1]
            span = b1 + 1 - first           # (every run of a valid k-mer clamped at 255: the < 256 test is the same, k_sketch.h)
            if span < 256:
                kf = 0
                for u in range(t - k + 1, t + 1):
                    kf = kf << 2 | st[u][0]
                kr = 0
                for u in range(t, t - k, -1):
                    kr = kr << 2 | (3 ^ st[u][0])
                assert kf != kr
                z = 0 if kf < kr else 1
                x = hash64(kr if z else kf, mask) << 8 | span
                y = b1 << 1 | z
        xs.append(x); ys.append(y); ls.append(l)
    return xs, ys, ls


def emitted_steps(xs, ls, w, k):
    """the position-parallel rule: the set of steps mm_sketch writes out, as a sorted list"""
    T = len(xs)
    X = lambda t: xs[t] if 0 <= t < T else MAXX

    def m(t):       # right-most smallest of steps t-w+1 .. t
        best, bx = t, X(t)
        for q in range(t - 1, t - w, -1):
            if X(q) < bx:
                best, bx = q, X(q)
        return best

    out = []
    for p in range(T):
        xp = xs[p]
        if xp == MAXX:
            continue
        emit = False
        # (A) p is the running minimum at some time (taken at step p, or by the rescan when a smaller one in front of it slides out), and
        #     leaves that role written.  E = the first step behind p whose x is not larger (it takes the minimum over), at most p + w
        #     (p slides out); p is the minimum at some time iff it is the right-most smallest of the LAST window it sits in before E.
        dE = next((d for d in range(1, w + 1) if p + d < T and xs[p + d] <= xp), None)
        t_last = min(p + (dE if dE is not None else w + 1) - 1, p + w - 1, T - 1)
        if all(X(q) >= xp for q in range(t_last - w + 1, p)):
            if dE is not None:
                emit = ls[p + dE] >= w + k            # replaced by a new element that is not larger
            elif p + w < T:
                emit = ls[p + w] >= w + k - 1         # slid out of the window
            else:
                emit = True                           # still the minimum at the end of the read: final flush
        # (B) the flush of equal minima at the first full window
        if not emit:
            for t0 in range(p + 1, min(p + w, T)):
                if ls[t0] == w + k - 1:
                    q = m(t0 - 1)
                    if X(q) != MAXX and X(q) == xp and q != p:
                        emit = True
        # (C) the flush of equal minima behind a rescan
        if not emit:
            for t in range(p + 1, min(p + w, T)):
                if t - w >= 0 and m(t - 1) == t - w and xs[t] > xs[t - w] and ls[t] >= w + k - 1:
                    q = m(t)
                    if X(q) != MAXX and X(q) == xp and q != p:
                        emit = True
        if emit:
            out.append(p)
    return out


def sketch(seq, w, k, rid=0, hpc=False):
    xs, ys, ls = infos_of(seq, w, k, hpc)
    return [(xs[p], rid << 32 | ys[p]) for p in emitted_steps(xs, ls, w, k)]
