#!/usr/bin/env python
"""A/B of environment knobs through bench.py inside ONE gpurun call (a `bench.py --no-cpu-baseline` run costs 3.5-7 s on the box once
torch is paged in; boxes differ by up to 10 %, so only numbers of one call compare).  The variants are run round-robin `--rounds`
times, so drift of the box shows up as spread inside every variant instead of as a difference between them.

  python tools/ab_bench.py --tag r03a base: mfma32:OCR_HALO_MFMA32=1 "old:OCR_W9_DEFER=0,OCR_FUSE_PACK_BIAS=0"
  python tools/ab_bench.py --tag r03deep --bench-args "--workload deep" base: nodefer:OCR_W9_DEFER=0

Each spec is label:ENV=VALUE[,ENV=VALUE...] (an empty list = the defaults).  Prints one line per variant (mean / min / max ms per step,
images/s of the mean, the convolution roofline number) and writes gpurun_out/<tag>_ab.json.
"""
import argparse
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def run_once(env_add, bench_args, steps, timeout):
    env = dict(os.environ)
    env.update(env_add)
    cmd = [sys.executable, os.path.join(ROOT, 'bench.py'), '--no-cpu-baseline', '--steps', str(steps)] + bench_args
    try:
        out = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=timeout, cwd=ROOT)
    except subprocess.TimeoutExpired:
        return None
    for line in reversed(out.stdout.strip().splitlines()):
        if line.startswith('{'):
            try:
                return json.loads(line)
            except ValueError:
                pass
    return None


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('specs', nargs='+', help='label:ENV=VALUE,ENV=VALUE ...')
    ap.add_argument('--tag', default='ab')
    ap.add_argument('--rounds', type=int, default=2)
    ap.add_argument('--steps', type=int, default=300)
    ap.add_argument('--timeout', type=int, default=120)
    ap.add_argument('--bench-args', default='', help='extra arguments for bench.py, e.g. "--workload deep"')
    args = ap.parse_args()
    variants = []
    for spec in args.specs:
        label, _, envs = spec.partition(':')
        env = dict(kv.split('=', 1) for kv in envs.split(',') if kv)
        variants.append((label, env))
    bench_args = args.bench_args.split()
    results = {label: [] for label, _ in variants}
    for _ in range(args.rounds):
        for label, env in variants:
            d = run_once(env, bench_args, args.steps, args.timeout)
            results[label].append(None if d is None else dict(ms_per_step=d['ms_per_step'], images_per_s=d['value'],
                                                              conv_tflops=d.get('roofline', {}).get('achieved')))
    table = []
    for label, env in variants:
        ok = [r for r in results[label] if r]
        if not ok:
            print('%-16s no result' % label)
            table.append(dict(label=label, env=env, runs=results[label]))
            continue
        ms = [r['ms_per_step'] for r in ok]
        mean = sum(ms) / len(ms)
        batch = ok[0]['images_per_s'] * ok[0]['ms_per_step'] / 1e3
        print('%-16s %.4f ms (min %.4f max %.4f, %d runs)  %7.0f images/s  conv %s TFLOP/s'
              % (label, mean, min(ms), max(ms), len(ms), batch / mean * 1e3,
                 '/'.join('%.0f' % r['conv_tflops'] for r in ok if r['conv_tflops'])))
        table.append(dict(label=label, env=env, mean_ms_per_step=mean, min_ms_per_step=min(ms), max_ms_per_step=max(ms), runs=results[label]))
    out_dir = os.path.join(ROOT, 'gpurun_out')
    os.makedirs(out_dir, exist_ok=True)
    with open(os.path.join(out_dir, '%s_ab.json' % args.tag), 'w') as f:
        json.dump(dict(steps=args.steps, rounds=args.rounds, bench_args=bench_args, variants=table), f, indent=1)


if __name__ == '__main__':
    main()
