#!/bin/bash
# A/B an environment knob on the headline step: tools/ab_env.sh OUTDIR NAME=VALUE...  (each setting one short bench run, back to back on one box)
out=$1; shift
mkdir -p "$out"
Q="--no-cpu-baseline --no-roofline --no-fwd-leg --no-extras --steps 40 --warmup 8"
for kv in "$@"; do
  name=$(echo "$kv" | tr '= ' '__')
  env $kv timeout -k 10 200 python bench.py $Q > "$out/$name.json" 2> "$out/$name.err" || exit 1
  python -c "import json;d=json.load(open('$out/$name.json'));print('$kv',d['value'],d['ms_per_step'])"
done
