#!/bin/bash
# Not a test: round 4, first device session (through gpurun from the repo root): the whole -m gpu suite, the default bench line, the
# cache-policy experiment on the decision traffic (variant libraries under gpurun_in/), the copy denominator sweep.
cd /tmp && export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r4a; rm -rf $O; mkdir -p $O
timeout 900 python -m pytest tests -m gpu -x -q > $O/gputest.log 2>&1; echo "pytest rc $?" >> $O/gputest.log
tail -5 $O/gputest.log
timeout 400 python bench.py > $O/bench.json 2> $O/bench.err; echo "bench rc $?"
timeout 60 tools/ubench/copy_f4 > $O/copy_f4.txt 2>&1
for v in "" nt_store nt_load nt_both; do
  L=welle.io_amd/libdabphy_hip.so; [ -n "$v" ] && L=gpurun_in/lib_$v.so
  DABPHY_LIB=$PWD/$L timeout 200 python tools/time_fused.py >> $O/cache_policy_alone.txt 2>> $O/cache_policy.err
  DABPHY_LIB=$PWD/$L timeout 300 python bench.py --no-cpu-baseline --no-alt-schedule --no-extras --steps 20 > $O/bench_${v:-plain}.json 2>> $O/cache_policy.err
done
for fic in 1 0; do
  DABPHY_FUSED_FIC=$fic DABPHY_LIB=$PWD/gpurun_in/lib_exp.so timeout 300 python bench.py --no-cpu-baseline --no-alt-schedule --no-extras --steps 20 > $O/bench_fic$fic.json 2>> $O/cache_policy.err
  DABPHY_FUSED_FIC=$fic DABPHY_LIB=$PWD/gpurun_in/lib_exp.so timeout 200 python tools/time_fused.py >> $O/fic_alone.txt 2>> $O/cache_policy.err
done
python - <<'PY'
import json, glob, os
for f in sorted(glob.glob("gpurun_out/r4a/bench*.json")):
    try:
        j = json.loads([l for l in open(f) if l.startswith("{")][-1])
        print(os.path.basename(f), "ms_per_step %.3f value %.0f demod %.3f vit %.3f fic %.3f frac %.3f" % (j["ms_per_step"], j["value"], j["stages_ms"]["demod"], j["stages_ms"]["msc_viterbi"], j["stages_ms"]["fic"], j["roofline"]["frac"]),
              "hetero", (j.get("extras") or {}).get("hetero", {}).get("value"), (j.get("extras") or {}).get("hetero", {}).get("msc_viterbi_ms"), (j.get("extras") or {}).get("hetero", {}).get("parity"),
              "copy", j["roofline"].get("measured_copy_GBps"), "facade", (j.get("facade") or {}).get("ms_per_frame"))
    except Exception as e:
        print(f, "unreadable", e)
PY
cat $O/cache_policy_alone.txt $O/fic_alone.txt; tail -3 $O/copy_f4.txt; tail -5 $O/cache_policy.err
