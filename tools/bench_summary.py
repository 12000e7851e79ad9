#!/usr/bin/env python3
"""one line per bench JSON file: clips/s, ms per step, sampled clock / power, dominant-kernel fraction, all-conv TFLOP/s"""
import json
import sys

for f in sys.argv[1:]:
    try:
        line = [l for l in open(f).read().splitlines() if l.startswith("{")][-1]
        d = json.loads(line)
        r, c = d.get("roofline", {}), d.get("clocks", {})
        print(f"{f}: {d['value']:.1f} {d['unit']}  {d['ms_per_step']:.3f} ms/step over {d['steps']}  sclk {c.get('sclk_mhz_avg')} MHz {c.get('socket_power_w_avg')} W  "
              f"dominant {r.get('frac')} ({r.get('avg_launch_ms')} ms)  all-conv {r.get('all_conv_launches', {}).get('TFLOP/s')} TF  host {d.get('host_enqueue_ms_per_step')} ms")
    except Exception as e:                       # noqa: BLE001
        print(f"{f}: unreadable ({e})")
