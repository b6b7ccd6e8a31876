"""Sequential model of the list ranking as the kernels do it (k_rank_rulers with tags, k_rank_lds with terminal / dead /
hopping entries, k_link_jump resolving tags), checked against a direct walk on random link structures: chains of all
lengths, isolated cycles, dropped edges.  A logic check of the restatement in kernels_build.hip — not of its concurrency.
usage: python tools/rank_model.py [seeds]     (tests/test_rank_model.py runs a few seeds on the CPU)"""
import random
import sys

NONE32 = 0xFFFFFFFF
RANK_NONE = 0xFFFFFFFF
DEAD = 0x8000FFFF
DONE = 1 << 31
TAG = 1 << 30


def make(n, rng, p_cycle=0.15, p_dead=0.02):
    """link[i] = (ancestor, done, distance) or None, as k_edge_next leaves them; truth[i] = (head, distance) or None"""
    order = list(range(n))
    rng.shuffle(order)
    link, truth = [None] * n, [None] * n
    i = 0
    while i < n:
        ln = min(n - i, max(1, int(rng.expovariate(1 / rng.choice([3, 30, 300, 3000])))))
        ch = order[i:i + ln]
        i += ln
        if rng.random() < p_cycle:            # isolated cycle: everybody has a predecessor, nobody is a head
            for j, e in enumerate(ch):
                link[e] = (ch[j - 1], False, 1)
            continue
        dead_at = None
        for j, e in enumerate(ch):
            if j == 0:
                link[e] = (e, True, 0)
                truth[e] = (e, 0)
            else:
                link[e] = (ch[j - 1], ch[j - 1] == ch[0], 1)
                if dead_at is None and rng.random() < p_dead:
                    link[e] = None
                    dead_at = j
                truth[e] = (ch[0], j) if dead_at is None else None
    return link, truth


def rank_rulers(link, n, rshift):
    """k_rank_rulers: entry per ruler = (nearest ruler or head towards the head) << 16 | distance; the edges a walk passes
    get a tag (ruler, how far ahead of it they lie)"""
    rmask = (1 << rshift) - 1
    nr = (n + rmask) >> rshift
    ent = [RANK_NONE] * nr
    for r in range(nr):
        i = r << rshift
        cur, acc = i, 0
        for _ in range(4096):
            l = link[cur]
            if cur != i:
                link[cur] = ("tag", i, acc)
            if l is None:
                break
            a, done, d = l
            acc += d
            if done or not (a & rmask):
                ent[r] = (a << 16) | acc
                break
            if a == i:
                break
            cur = a
    return ent


def rank_lds(ent, rshift, max_rounds=18):
    """k_rank_lds: returns the rulers' final (head, distance) or None"""
    rmask = (1 << rshift) - 1
    nr = len(ent)
    s, live = [0] * (nr + 1), [False] * nr
    for r in range(nr):
        e = ent[r]
        if e == RANK_NONE:
            s[r] = DEAD
            continue
        a = e >> 16
        if (a & rmask) or (a == (r << rshift) and not (e & 0xFFFF)):
            s[r] = (0x8000 | r) << 16
        elif a == (r << rshift):
            s[r] = DEAD
        else:
            s[r] = ((a >> rshift) << 16) | (e & 0xFFFF)
            live[r] = True
    mine = list(s)
    for _ in range(max_rounds):
        hopped = False
        new = list(s)
        for r in range(nr):
            e = mine[r]
            x = s[(e >> 16) & 0x7FFF]
            stop = x >> 31
            hop = live[r] and not stop
            if hop:
                e = (x & 0xFFFF0000) | ((e + x) & 0xFFFF)
            if live[r] and stop:
                live[r] = False
            hopped = hopped or hop
            mine[r] = new[r] = e
        s = new
        if not hopped:
            break
    out = [None] * nr
    for r in range(nr):
        e = mine[r]
        if live[r] or e == DEAD:
            continue
        t = (e >> 16) & 0x7FFF
        if s[t] == DEAD:
            continue
        out[r] = (ent[t] >> 16, ((e & 0xFFFF) + (ent[t] & 0xFFFF)) & 0xFFFF)
    return out


def link_jump(link, final_rulers, n, rshift):
    """k_link_jump after the rulers' links are final: tagged edges take the ruler's link minus their tag, the others walk"""
    res = [None] * n
    for r, f in enumerate(final_rulers):
        res[r << rshift] = f
    rmask = (1 << rshift) - 1
    for i in range(n):
        if not (i & rmask):
            continue
        l = link[i]
        if l is None:
            continue
        if l[0] == "tag":
            f = res[l[1]]
            res[i] = (f[0], f[1] - l[2]) if f is not None else None
            continue
        cur, acc, ok = i, 0, False
        for _ in range(100000):
            l = link[cur]
            if l is None or l[0] == "tag":        # (a tagged link ahead of an untagged edge cannot be)
                break
            a, done, d = l
            acc += d
            if done:
                res[i] = (a, acc)
                ok = True
                break
            if not (a & rmask):
                f = res[a]
                if f is not None:
                    res[i] = (f[0], f[1] + acc)
                ok = True
                break
            if a == i:
                break
            cur = a
        del ok
    return res


def check(seed):
    rng = random.Random(seed)
    n = rng.choice([1, 2, 5, 17, 100, 1000, 5000, 20000, 40000])
    rshift = rng.choice([1, 1, 2, 3])
    link, truth = make(n, rng, p_dead=rng.choice([0.0, 0.002, 0.05]))
    ent = rank_rulers(link, n, rshift)
    res = link_jump(link, rank_lds(ent, rshift), n, rshift)
    return [(i, res[i], truth[i]) for i in range(n) if res[i] != truth[i]]


if __name__ == "__main__":
    bad = 0
    for seed in range(int(sys.argv[1]) if len(sys.argv) > 1 else 100):
        m = check(seed)
        bad += len(m)
        if m:
            print("seed", seed, "mismatches", m[:3])
    print("mismatches:", bad)
