#!/usr/bin/env python3
"""Reads the [trace] lines of a PAG_WALK_TRACE=1 run (stderr of bench.py / bin/pagraph; the LAST walk of the log) and prints where
the time of the walks went: how long the device was saturated, the last jobs with the lag between a job's end on the device and
the control thread seeing it, and per contig with more than one round: when each round was decided and what its chain jobs took.

usage: python tests/walk_trace.py LOG [waves]"""
import re
import sys


def main():
    lines = open(sys.argv[1]).read().split("\n")
    waves = int(sys.argv[2]) if len(sys.argv) > 2 else 1024
    starts = [i for i, ln in enumerate(lines) if ln.startswith("[trace] walks")]
    if not starts:
        sys.exit("no [trace] lines")
    body = lines[starts[-1]:]
    print(body[0])
    done, post, over, loop = [], [], [], []
    for ln in body[1:]:
        if not ln.startswith("[trace] "):
            break
        m = re.match(r"\[trace\] loop t=([\d.]+) (.*) jobs (\d+) decided (\d+) live (\d+)", ln)
        if m:
            loop.append((float(m.group(1)), m.group(2), int(m.group(3)), int(m.group(4)), int(m.group(5))))
            continue
        m = re.match(r"\[trace\] done t=([\d.]+) ctg (\d+) (seg|chain) (-?\d+) dev ([\d.]+)\.\.([\d.]+) len (\d+) classify (\d+)", ln)
        if m:
            done.append(dict(t=float(m.group(1)), ctg=int(m.group(2)), kind=m.group(3), idx=int(m.group(4)), b=float(m.group(5)), e=float(m.group(6)),
                             len=int(m.group(7)), cls=int(m.group(8))))
            continue
        m = re.match(r"\[trace\] post t=([\d.]+) ctg (\d+) (seg|chain) (-?\d+) ring (\d+) mode (\d+) init (\d+)", ln)
        if m:
            post.append(dict(t=float(m.group(1)), ctg=int(m.group(2)), kind=m.group(3), idx=int(m.group(4)), ring=int(m.group(5)), mode=int(m.group(6)), init=int(m.group(7))))
            continue
        m = re.match(r"\[trace\] over t=([\d.]+) ctg (\d+) round (\d+) chain (-?\d+) len (\d+)( leap)?", ln)
        if m:
            over.append(dict(t=float(m.group(1)), ctg=int(m.group(2)), round=int(m.group(3)), len=int(m.group(5)), leap=bool(m.group(6))))
    off = min(d["t"] - d["e"] for d in done)  # host clock of the device's zero, as the least lag says
    print(f"{len(done)} jobs, {len(post)} postings; device zero at host t = {off:.2f} ms; last device end {max(d['e'] for d in done) + off:.2f} ms (host clock)")
    # jobs running per 5 ms of device time
    end = max(d["e"] for d in done)
    prof = []
    for lo in range(0, int(end) + 5, 5):
        busy = sum(max(0.0, min(d["e"], lo + 5) - max(d["b"], lo)) for d in done) / 5.0
        prof.append(f"{busy:.0f}")
    print("jobs running, per 5 ms of device time:", " ".join(prof))
    busy_total = sum(d["e"] - d["b"] for d in done)
    print(f"wave time {busy_total:.0f} ms = {busy_total / waves:.1f} ms on each of {waves} waves")
    lags = sorted(d["t"] - off - d["e"] for d in done)
    print(f"lag device end -> control thread: median {lags[len(lags) // 2]:.2f} ms, 90 % {lags[len(lags) * 9 // 10]:.2f}, max {lags[-1]:.2f}")
    print("last jobs (host clock):")
    for d in sorted(done, key=lambda d: -d["e"])[:14]:
        print(f"  ctg {d['ctg']:3d} {d['kind']:5s} {d['idx']:4d} ran {d['b'] + off:7.2f} .. {d['e'] + off:7.2f} ({d['e'] - d['b']:5.2f} ms, {d['len']} vertices, {d['cls']} classifications), seen {d['t']:7.2f}")
    multi = sorted({o["ctg"] for o in over if o["round"] > 1})
    print("contigs with more than one round:", multi)
    for c in multi:
        print(f" contig {c}:")
        ev = [("over", o["t"], f"round {o['round']} decided, {o['len']} vertices" + (", leap" if o["leap"] else "")) for o in over if o["ctg"] == c]
        ev += [("post", p["t"], f"posted chain {p['idx']} ring {p['ring']} mode {p['mode']} init {p['init']}") for p in post if p["ctg"] == c and p["kind"] == "chain"]
        ev += [("done", d["t"], f"chain {d['idx']} ran {d['b'] + off:.2f}..{d['e'] + off:.2f} ({d['e'] - d['b']:.2f} ms, +{d['len']} vertices, {d['cls']} classifications)")
               for d in done if d["ctg"] == c and d["kind"] == "chain"]
        last_round1 = min((o["t"] for o in over if o["ctg"] == c), default=0)
        for what, t, txt in sorted(ev, key=lambda e: e[1]):
            if t >= last_round1 - 0.01:
                print(f"   t={t:7.2f} {txt}")
    # the control thread's laps once the device has run out of work
    t_sat = next((lo for lo in range(0, int(end) + 5, 5) if lo > 20 and sum(max(0.0, min(d["e"], lo + 5) - max(d["b"], lo)) for d in done) / 5.0 < waves / 2), end) + off
    print(f"control thread after t = {t_sat:.0f} ms (fewer than half the waves busy): lap ends, what, jobs / decided / live")
    prev = None
    agg = {}
    for t, what, nj, nd, live in loop:
        if prev is not None and t >= t_sat:
            agg[what] = agg.get(what, 0.0) + (t - prev)
            if t - prev >= 0.5:
                print(f"   t={t:7.2f} {what:20s} took {t - prev:5.2f} ms ({nj} jobs, {nd} decided, {live} live)")
        prev = t
    print("   sums:", ", ".join(f"{k} {v:.1f}" for k, v in sorted(agg.items(), key=lambda kv: -kv[1])))
    print("decisions:", " ".join(f"{o['ctg']}:{o['t']:.0f}" for o in sorted(over, key=lambda o: o["t"])))


if __name__ == "__main__":
    main()
