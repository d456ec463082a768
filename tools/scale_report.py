#!/usr/bin/env python
"""Weak-scaling table from bench.py result lines (the driver's SCALE_rNN.json / BENCH_rNN.json, or files of JSON lines):
samples/s, ms/step, speed-up and efficiency against the N = 1 line, next to the communication-free prediction from the
one-GPU emulation (per-rank compute of an N-rank job, replicated vs row-sharded tables).

    python tools/scale_report.py SCALE_r01.json [more files ...]
"""
import json
import sys

# per-rank ms/step measured on one GPU with RECHUB_EMULATE_WORLD=N (no wire time), tools/shard_bench.sh, round 1
EMULATED_MS = {"shard": {1: 0.425, 2: 0.364, 8: 0.385}, "replicate": {1: 0.411, 2: 0.426, 4: 0.493, 8: 0.557}}


def lines_of(path):
    text = open(path).read().strip()
    try:
        doc = json.loads(text)
    except json.JSONDecodeError:
        return [json.loads(l) for l in text.splitlines() if l.lstrip().startswith("{")]
    if isinstance(doc, dict) and "metric" in doc:
        return [doc]
    found = []

    def walk(o):  # the driver may nest the lines (per-N entries, lists, "result" keys ...)
        if isinstance(o, dict):
            if "metric" in o and "n_gpus" in o:
                found.append(o)
            else:
                for v in o.values():
                    walk(v)
        elif isinstance(o, list):
            for v in o:
                walk(v)

    walk(doc)
    return found


def main():
    rows = [r for p in sys.argv[1:] for r in lines_of(p)]
    if not rows:
        print(__doc__)
        return 1
    rows.sort(key=lambda r: r["n_gpus"])
    base = next((r for r in rows if r["n_gpus"] == 1), rows[0])
    print(f"{'N':>2} {'samples/s':>14} {'ms/step':>8} {'speed-up':>9} {'efficiency':>10}  tables / graph / no-wire bound")
    for r in rows:
        n = r["n_gpus"]
        speed = r["value"] / base["value"] * base["n_gpus"]
        cfg = r.get("config", {})
        placement = "shard" if "shard" in str(cfg.get("tables", "")) else "replicate"
        emu = EMULATED_MS[placement].get(n)
        bound = f"{base['ms_per_step'] / emu * n:.2f}x" if emu else "-"
        print(f"{n:>2} {r['value']:>14,.0f} {r['ms_per_step']:>8.4f} {speed:>8.2f}x {speed / n:>9.1%}  "
              f"{placement} / {'hipGraph' if cfg.get('hipgraph') else 'eager'} / {bound}")
    return 0


if __name__ == "__main__":
    sys.exit(main())
