"""TEST INFRASTRUCTURE ONLY -- CPU restatement (numpy, float64) of the reference's coverage model fit, row N4.

`ModelDist HISTO K ReadLength Threads` (reference `src/ModelDist.cpp`, called at `runRufus.sh:849`) reads a
tab-separated `jellyfish histo -f` table and fits, by a coordinate-wise 11-point grid search, a sum of normal
curves (half-copy, 1x, 2x, ... copies) on top of a 1/x^p error curve; `runRufus.sh:862-868` takes lines 2 and 4
of `HISTO.7.7.model` as MutantMinCov and MutantSC.

Only `tests/` may import this module.  It is pinned by the reference binary itself: `oracle/_ref/ModelDist`
(compiled from `/root/reference/src/ModelDist.cpp` by `make -C oracle ref`) was run on the fixtures under
`tests/golden/modeldist/` by `tests/golden/make_golden_modeldist.py`, and `tests/test_modeldist.py` compares
this restatement with those outputs token by token.

Things the reference does that are kept: pi = 3.14159; the histogram vector starts at the first row with a
non-zero count, so index i is NOT the multiplicity when row 1 is empty (`:437-452`); the last copy-number column
is never normalised (`:107-117`: the loops over j stop one short of the diploid layout); `float` accumulators in
the error fit and the cutoff (`:344-358`, `:836-853`); a `long` accumulator for line 5 of the model file
(`:885-893`); in main() `prob` is shifted by one row against `dist` (`:684`, `:760-772`).
Undefined behaviour that is NOT emulated (the values the reference prints there are whatever the heap holds):
`values[9]` written at [9] and [10] (`:546`) -- eleven values are kept; `prob[SC*a][a+1]` for the last a in main()
reads one past the row (`:805`) -- 0 is used; `ErrorDist[n]` in the last row of the .prob file (`:955`) -- 0.
"""
import math
import numpy as np

PI = 3.14159  # :28


def _g(x):
    """`ostream << double` at the default precision."""
    x = float(x)
    if math.isnan(x):
        return "nan"
    return "%g" % x


def _norm(x, mu, sigma, skew, p):
    """:31-37, vectorised over x (rows) and mu/sigma (columns)."""
    with np.errstate(all="ignore"):
        if skew == 0 and p > 0:
            s = np.broadcast_to(sigma, np.broadcast(x, mu).shape)  # pow(+-0, p) = 0: sigma is left as it is
        else:
            below = x < mu
            s = np.where(below, sigma + np.power(np.where(below, (mu - x) * skew, 0.0), p), sigma)
        arg = ((x - mu) ** 2) / (2 * s * s)
        e = np.zeros(arg.shape)
        near = ~(arg >= 746)        # exp(-746) is 0 in binary64: only the band around mu (and NaNs) is evaluated
        e[near] = np.exp(-arg[near])
        return (1.0 / np.sqrt(2 * PI * s * s)) * e


def _columns(n, SC, stdev, factor):
    """mu and sigma of columns 1..Jn+1 (:89-99): the half-copy column, then j = 1..Jn copies, j < n / SC."""
    q = n / SC
    Jn = int(math.ceil(q)) - 1
    j = np.arange(1, Jn + 1, dtype=np.float64)
    mu = np.concatenate(([SC / 2], SC * j))
    with np.errstate(all="ignore"):
        het = stdev * (1 - ((1 - (stdev / (stdev + (1 * factor)))) / 2))
    sigma = np.concatenate(([het], stdev + ((j - 1) * factor)))
    return Jn, mu, sigma


def _tables(hist, SC, stdev, factor, skew, power, i0):
    """dist (n x (Jn+2), column 0 zero), row totals over columns 1..Jn: :84-128 (i0 = 1) and :716-772 (i0 = 0)."""
    n = len(hist)
    Jn, mu, sigma = _columns(n, SC, stdev, factor)
    x = np.arange(n, dtype=np.float64)[:, None]
    D = np.zeros((n, Jn + 2))
    D[i0:, 1:] = _norm(x[i0:], mu[None, :], sigma[None, :], skew, power)
    with np.errstate(all="ignore"):
        colsum = D[i0:, 1:Jn + 1].sum(axis=0)
        D[i0:, 1:Jn + 1] = D[i0:, 1:Jn + 1] / colsum
        rowtot = D[:, 1:Jn + 1].sum(axis=1)
    return Jn, D, rowtot


def _rc(hist, D, rowtot, Jn, SC, prob_shift):
    """:130-163 / :774-812.  prob_shift = 1 in main(): prob[r] there is the row r - 1 of dist."""
    with np.errstate(all="ignore"):
        rc = np.zeros(Jn + 2)
        s, h = int(SC), int(SC / 2)
        tSC = hist[s] / D[s, 2]
        het = (hist[h] - (D[h, 2] * tSC)) / D[h, 1]
        rc[1] = het if het > 0 else 0.0
        rc[2] = hist[s] / D[s, 2]
        for a in range(2, Jn + 1):
            r = int(SC * a)
            if prob_shift and a == Jn:
                p = 0.0  # (the reference reads past the end of the row here)
            else:
                p = D[r - prob_shift, a + 1] / rowtot[r - prob_shift]
            rc[a + 1] = float(hist[r]) / D[r, a + 1] * p
    return rc


def _test_model(log, SC, stdev, factor, skew, power, hist2, inflection, max_copy):
    """testModelLog (:72-198) / testModel (:200-318): the residual the grid search minimises."""
    Jn, D, rowtot = _tables(hist2, SC, stdev, factor, skew, power, 1)
    rc = _rc(hist2, D, rowtot, Jn, SC, 0)
    with np.errstate(all="ignore"):
        idx = []
        i = float(inflection)
        while i < SC * max_copy:
            idx.append(int(i))
            i += 1
        idx = np.array(idx, dtype=np.int64)
        hv = hist2[idx].astype(np.float64)
        mv = (D[idx, 1:Jn + 1] * rc[1:Jn + 1]).sum(axis=1)
        if log:
            t = (np.log(hv) - np.log(mv)) ** 2
        else:
            t = (hv - mv) ** 2
        acc = 0.0
        for v in t:  # in the reference's order: a NaN or an infinity must appear where it does there
            acc += v
    return acc


def _grid(low, high, f):
    return [f(low + ((high - low) / 10) * x) for x in range(11)]


def _argmin(values):
    lowest, at = values[0], 0
    for x in range(1, 11):
        if values[x] < lowest:
            lowest, at = values[x], x
    return at


def _fit_error_model(E, mx):
    """:339-365 (float p, float bestP)."""
    with np.errstate(all="ignore"):
        i = np.arange(1, max(mx, 1), dtype=np.float64)
        e = E[1:max(mx, 1)]
        last = float(np.sum((np.log(e) - np.log((1 / np.power(i, 100.0)) * E[1])) ** 2)) if len(i) else 0.0
        best = np.float32(0)
        p = np.float32(7)
        while float(p) > .1:
            pd = float(p)
            ssq = 0.0
            for ii, ee in zip(i, e):
                ssq += _sq(_log(ee) - _log((1 / math.pow(ii, pd)) * E[1]))
            if ssq < last:
                last, best = ssq, p
            p = np.float32(float(p) + -.001)
        bp = float(best)
        k = np.arange(len(E), dtype=np.float64)
        out = E.copy()
        out[1:] = (1 / np.power(k[1:], bp)) * E[1]
        total = 0.0
        for v in out[1:]:
            total += v
    return out, total, best


def _log(v):
    if v > 0:
        return math.log(v)
    if v == 0:
        return -math.inf
    return math.nan


def _sq(v):
    return v * v


def model_dist(histo_text, hs, read_length, threads=1, name="HISTO"):
    """Returns (exit_code, stdout, {".7.7.model": text, ".7.7.dist": text, ".7.7.prob": text})."""
    out = []
    P = out.append
    P("Call is histoFile HS ReadLength Threads")
    P(f"Parent File open - {name}")
    lines = histo_text.split("\n")
    if lines and lines[-1] == "":
        lines.pop()
    pos = 0

    def atoi(s):
        s = s.strip()
        m = 0
        sign = 1
        j = 0
        if j < len(s) and s[j] in "+-":
            sign = -1 if s[j] == "-" else 1
            j += 1
        while j < len(s) and s[j].isdigit():
            m = m * 10 + int(s[j])
            j += 1
        return sign * m

    temp = lines[pos].split("\t"); pos += 1
    P(f"first line = {temp[0]} - {temp[1]}")
    count = 0
    while atoi(temp[1]) == 0 or atoi(temp[0]) == 0:
        P("getting another ")
        temp = lines[pos].split("\t"); pos += 1
        P(f"got {temp[0]} - {temp[1]}")
        count += 1
        if count > 10:
            P("ERROR there are no kmers in this file")
            return 1, "\n".join(out) + "\n", {}
    P(f"going with {temp[0]} - {temp[1]}")
    value = float(atoi(temp[1]))
    histo = [0, int(value)]
    last = value
    past = False
    i = 1
    total = 0
    total_kmers = 0
    histo_sum = 0.0
    SC, SCvalue, inflection = 1.0, -1.0, -1
    while pos < len(lines):
        temp = lines[pos].split("\t"); pos += 1
        i += 1
        value = float(atoi(temp[1]))
        histo.append(int(value))
        total += int(value)
        total_kmers += int(value) * atoi(temp[0])
        histo_sum += value
        if value - last > 0 and not past:
            inflection = i - 1
            past = True
        if past and SCvalue < value:
            SCvalue = value
            SC = float(i)
        last = float(histo[i])
    n = len(histo)
    hist = np.array(histo, dtype=np.int64)
    P(f"Number of reads = {int(total_kmers / (read_length - hs + 1)) if total_kmers >= 0 else 0}")
    ybar = total / i
    for j in range(10):
        P(f"I = {j} \t {histo[j]}")
    P(f"SC = {_g(SC)} vlaue = {_g(SCvalue)}")
    raw_sc = int(SC)
    stdvalue = SCvalue * math.exp(-.5)
    j = int(SC)
    while j < n:
        if histo[j] - stdvalue < 0:
            break
        j += 1
    stdev = float(j - SC)
    P(f"stdi = {j} stdev = {_g(stdev)}")
    E = hist.astype(np.float64)
    E, burner, best_p_err = _fit_error_model(E, inflection)
    P(f"best error is 1/x^{_g(np.float32(best_p_err))}")
    with np.errstate(all="ignore"):
        error_dist = E / burner
    diff = hist - E
    hist2 = np.where(diff > 0, np.trunc(diff), 0).astype(np.int64)

    factor, skew, power = 1.0, 0.0, 1.0
    bestS, bestF, bestSC, bestSK, bestP = stdev, factor, SC, skew, power
    ev = lambda log, sc, st, f, sk, p: _test_model(log, sc, st, f, sk, p, hist2, inflection, 5)
    for pas in range(3):
        P(f"On {pas + 1} pass")
        lo, hi, steps = 1.0, 20.0, 0
        while lo / hi < .999 and hi > 1e-10:
            steps += 1
            at = _argmin(_grid(lo, hi, lambda v: ev(True, bestSC, bestS, v, bestSK, bestP)))
            lo = lo + ((hi - lo) / 10) * (at - 1) if lo + ((hi - lo) / 10) * (at - 1) >= 0 else 0.0
            hi = lo + ((hi - lo) / 10) * (at + 1)
            bestF = lo + ((hi - lo) / 10) * at
        P(f"\t best Factor = {_g(bestF)} steps = {steps}")
        lo, hi, steps = SC * .9, SC * 1.1, 0
        while lo / hi < .999 and hi > 1e-50:
            steps += 1
            at = _argmin(_grid(lo, hi, lambda v: ev(False, v, bestS, bestF, bestSK, bestP)))
            lo = lo + ((hi - lo) / 10) * (at - 1) if lo + ((hi - lo) / 10) * (at - 1) >= 0 else 0.0
            hi = lo + ((hi - lo) / 10) * (at + 1)
            bestSC = lo + ((hi - lo) / 10) * at
        P(f"\t\tbestSC = {_g(bestSC)} steps = {steps}")
        lo, hi, steps = stdev * .9, stdev * 1.1, 0
        while lo / hi < .99 and hi > 1e-50:
            steps += 1
            at = _argmin(_grid(lo, hi, lambda v: ev(False, bestSC, v, bestF, bestSK, bestP)))
            lo = lo + ((hi - lo) / 10) * (at - 1) if lo + ((hi - lo) / 10) * (at - 1) >= 0 else 0.0
            hi = lo + ((hi - lo) / 10) * (at + 1)
            bestS = lo + ((hi - lo) / 10) * at
        P(f"\t\tbest StdDev = {_g(bestS)} steps = {steps}")
        # :605 -- `SKhigh < 1e-50` with SKhigh = 2: the skew search never runs
        P(f"\t\tbest skew factor = {_g(bestSK)} steps = 0")
        lo, hi, steps = 1.0, 2.0, 0
        while lo / hi < .999 and hi > 1e-50:
            steps += 1
            at = _argmin(_grid(lo, hi, lambda v: ev(True, bestSC, bestS, bestF, bestSK, v)))
            lo = lo + ((hi - lo) / 10) * (at - 1) if lo + ((hi - lo) / 10) * (at - 1) >= 1 else 1.0
            hi = lo + ((hi - lo) / 10) * (at + 1)
            bestP = lo + ((hi - lo) / 10) * at
        P(f"\t\tbest Power factor = {_g(bestP)} steps = {steps}")
        stdev, factor, SC, skew, power = bestS, bestF, bestSC, bestSK, bestP
    P(f"Best Model is SC = {_g(SC)} StdDev = {_g(stdev)} F = {_g(factor)} skew = {_g(skew)} bestP = {_g(power)}")

    Jn, D, rowtot = _tables(hist, SC, stdev, factor, skew, power, 0)
    rc = _rc(hist, D, rowtot, Jn, SC, 1)
    C = Jn + 1
    with np.errstate(all="ignore"):
        model = np.zeros((n, Jn + 1))
        model[:, 1:] = D[:, 1:Jn + 1] * rc[1:Jn + 1]
        sums = np.zeros(n)
        for jj in range(1, Jn + 1):
            sums = sums + model[:, jj]
        prob = np.zeros((n, Jn + 1))
        prob[:, 1:] = D[:, 1:Jn + 1] / rowtot[:, None]
    genome = 0.0
    for jj in range(1, C + 1):
        genome += rc[jj] * jj
    P(f"GenomeSize = {_g(genome)}")
    kcutoff = -1
    for k in range(1, n):
        num = np.float32(0)
        for c in range(1, C + 1):
            num = np.float32(float(num) + D[k, c])
        with np.errstate(all="ignore"):
            v = float(num) / (float(num) + error_dist[k])
        P(f"prob not error = {_g(v)}")
        if v > 0.5:
            kcutoff = k
            P("this one")
            break
    P("here1")
    P("here")

    head = f"3\n{kcutoff}\n{_g(histo_sum)}\n{raw_sc}\n"
    m = [head]
    local = np.zeros(Jn + 1, dtype=np.int64)
    for k in range(1, n):  # a long accumulator: the sum is cut to an integer after every row
        with np.errstate(all="ignore"):
            local = np.trunc(local.astype(np.float64) + np.nan_to_num(model[k], nan=0.0, posinf=0.0, neginf=0.0)).astype(np.int64)
    m.append("".join(_g(float(local[c]) / histo_sum) + "\t" for c in range(1, Jn + 1)) + "\n")
    m.append("K\tRawCount\tErrorModel\tContSubtract\tModelSum\t1x\t2x\t3x\t4x\t5x\t6x\t7x\t8x\t9x\n")
    m.append("0\t0\t0\t0\t0" + "".join("\t" + _g(model[0, c]) for c in range(1, 10)) + "\n")
    k = 1
    while k < SC * 5:
        m.append(f"{k}\t{histo[k]}\t{_g(E[k])}\t0\t{_g(sums[k])}" + "".join("\t" + _g(model[k, c]) for c in range(1, 10)) + "\n")
        k += 1
    d = [head, _g(burner) + "\t0\t" + "".join(_g(rc[jj]) + "\t" for jj in range(1, C + 1)) + "\n", _g(SC) + "\n"]
    row0 = "".join("\t" + _g(D[0, c]) for c in range(1, C + 1))
    d.append("0\t0\t0" + row0 + "\n")
    for k in range(1, n):
        d.append(f"{k}\t{_g(error_dist[k])}\t0" + "".join("\t" + _g(D[k, c]) for c in range(1, C + 1)) + "\n")
    d.append(row0)
    p = [head, "\n\n"]
    for k in range(1, n + 1):
        ed = error_dist[k] if k < n else 0.0
        p.append(f"{k}\t{_g(ed)}\t0" + "".join("\t" + _g(prob[k - 1, c]) for c in range(1, Jn + 1)) + "\n")
    P(f"GenomeSize = {_g(genome)}")
    P(f"Inflection point = {inflection}")
    P(f"Recomended RUFUS cutoff = {_g(SC - (5 * stdev))}")
    P(f"-1std = {_g(SC - stdev)}\t-2std = {_g(SC - 2 * stdev)}\t-3std = {_g(SC - 3 * stdev)}\t-4std = {_g(SC - 4 * stdev)}")
    return 0, "\n".join(out) + "\n", {".7.7.model": "".join(m), ".7.7.dist": "".join(d), ".7.7.prob": "".join(p)}
