#!/usr/bin/env python3
"""Summarise rocprofv3 output (rocpd sqlite .db) written by tools/profile.sh: per-kernel
durations from the kernel trace and per-dispatch averages of every PMC counter."""
import collections
import glob
import json
import os
import sqlite3
import sys


def dbs(d):
    return sorted(glob.glob(os.path.join(d, "**", "*.db"), recursive=True))


def kernel_stats(d):
    out = []
    for db in dbs(d):
        cur = sqlite3.connect(db).cursor()
        agg = collections.defaultdict(list)
        for name, dur in cur.execute("select name, duration from kernels"):
            agg[name].append(dur)
        for k, v in sorted(agg.items(), key=lambda kv: -sum(kv[1])):
            v2 = sorted(v)
            out.append(dict(kernel=k[:90], calls=len(v), total_us=round(sum(v) / 1e3, 1), avg_us=round(sum(v) / len(v) / 1e3, 2),
                            med_us=round(v2[len(v2) // 2] / 1e3, 2), min_us=round(v2[0] / 1e3, 2), max_us=round(v2[-1] / 1e3, 2)))
    return out


def window_stats(d, want, window):
    """The dominant kernel's dispatches split in time: the last `window` dispatches of the process (= bench.py's W warm-up + K
    timed steps, which run after the pre-heat loop's final synchronize) against everything before (the pre-heat phase), so
    that the average the profile reports and the bench line's ms_per_step describe the SAME launches."""
    out = []
    for db in dbs(d):
        cur = sqlite3.connect(db).cursor()
        try:
            rows = list(cur.execute("select start, duration from kernels where name like ? order by start", (f"%{want}%",)))
        except sqlite3.Error:
            try:
                rows = list(cur.execute("select start_timestamp, duration from kernels where name like ? order by start_timestamp", (f"%{want}%",)))
            except sqlite3.Error:
                continue
        if len(rows) <= window:
            continue
        durs = [r[1] for r in rows]
        pre, win = durs[:-window], durs[-window:]
        late = pre[len(pre) // 2:]
        med = lambda v: sorted(v)[len(v) // 2]       # noqa: E731
        out.append(dict(kernel=want, dispatches=len(durs), window=window,
                        timed_window_avg_us=round(sum(win) / len(win) / 1e3, 2), timed_window_med_us=round(med(win) / 1e3, 2),
                        preheat_avg_us=round(sum(pre) / len(pre) / 1e3, 2),
                        preheat_second_half_avg_us=round(sum(late) / len(late) / 1e3, 2), preheat_second_half_med_us=round(med(late) / 1e3, 2)))
    return out


def pmc_stats(d, want="pbl_gemv"):
    res = {}
    for db in dbs(d):
        cur = sqlite3.connect(db).cursor()
        agg = collections.defaultdict(list)
        dur = []
        try:
            rows = cur.execute("select counter_name, value, duration from counters_collection where kernel_name like ?",
                               (f"%{want}%",))
        except sqlite3.Error:
            continue
        for name, val, d_ in rows:
            agg[name].append(val)
            dur.append(d_)
        if agg:
            res = {k: round(sum(v) / len(v), 1) for k, v in agg.items()}
            res["_dispatches"] = len(dur) // len(agg)
            res["_avg_kernel_us_in_this_pass"] = round(sum(dur) / len(dur) / 1e3, 2)
    return res


def main():
    root = sys.argv[1]
    print(f"# rocprofv3 summary for {root}")
    tr = os.path.join(root, "trace")
    if os.path.isdir(tr):
        print("\n## kernel trace, un-instrumented timing (durations in us)")
        for s in kernel_stats(tr)[:6]:
            print(json.dumps(s))
        window = int(os.environ.get("PBL_PROF_WINDOW", "0"))     # W + K of the profiled command (tools/profile.sh exports it)
        if window:
            print("\n## the same trace split in time: last W + K dispatches (the timed region) vs the pre-heat loop")
            for s in window_stats(tr, "pbl_gemv", window):
                print(json.dumps(s))
    for sub in sorted(os.listdir(root)):
        p = os.path.join(root, sub)
        if not os.path.isdir(p):
            continue
        if sub.endswith("_trace"):
            print(f"\n## {sub}: kernel trace, un-instrumented timing (durations in us)")
            for s in kernel_stats(p)[:10]:
                print(json.dumps(s))
        want = "pbl_sb_img_kernel" if sub.startswith("sbimg") else "pbl_mfma_kernel" if sub.startswith("mfma") else ("pbl_gemm_img_kernel" if sub.startswith("gemmimg") else
                                                                ("pbl_gemm_kernel" if sub.startswith("gemm") else "pbl_gemv"))
        if "pmc" in sub or "fetch" in sub or "write" in sub or "tcc" in sub:
            st = pmc_stats(p, want)
            if st:
                print(f"\n## {sub}: per-dispatch averages for {want}")
                print(json.dumps(st))
    cal = os.path.join(root, "calib")
    if os.path.isdir(cal):
        st = pmc_stats(cal, "calib_stream_read")
        if st.get("FETCH_SIZE"):
            true_b = 2 * (1 << 30)
            f = true_b / (st["FETCH_SIZE"] * 1024.0)
            print(f"\n## FETCH_SIZE calibration: 2 GiB dwordx4 nt stream read reports FETCH_SIZE={st['FETCH_SIZE']:.0f} KiB "
                  f"-> correction factor {f:.3f}")
            p3 = pmc_stats(os.path.join(root, "pmc3"))
            if p3.get("FETCH_SIZE"):
                print(f"   corrected HBM read traffic of pbl_gemv_kernel: {p3['FETCH_SIZE'] * 1024 * f / 1e6:.1f} MB per dispatch")
    for log in sorted(glob.glob(os.path.join(root, "*.log"))):
        for line in open(log):
            if line.startswith("{\"metric\""):
                j = json.loads(line)
                r = j["roofline"]
                print(f"\n## bench line under {os.path.basename(log)}: value={j['value']:.0f} achieved={r['achieved']:.0f} {r['unit']} "
                      + (f"us/launch={r['us_per_launch']:.1f}" if "us_per_launch" in r else f"us/layer={r.get('us_per_layer', 0):.2f}"))


if __name__ == "__main__":
    main()
