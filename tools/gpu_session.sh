#!/bin/bash
# Not a test: ONE parameterised device session (replaces the per-session scripts r04_gpu_run*.sh / r05_gpu_run*.sh of earlier rounds).
#   gpurun --timeout T -- 'bash tools/gpu_session.sh <tag> "<command 1>" "<command 2>" ...'
# Every command runs from the repository root under its own `timeout` (first word "T=<seconds>" overrides the default 600), its output goes
# to gpurun_out/<tag>/<nn>.log, and the last lines of every log are echoed so that they are in gpurun's own tail.
# Shorthands: TESTS[:<-k expression>]  = python -m pytest tests -m gpu -x -q [-k ...]
#             BENCH[:<extra args>]     = python bench.py [...]  -> <nn>.json / <nn>.err
#             PROFILES                 = tools/make_profiles.sh (rocprofv3 passes of the default bench command; summaries under gpurun_out/prof)
cd /tmp && export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT" || exit 1
TAG=$1; shift
O=gpurun_out/$TAG; rm -rf "$O"; mkdir -p "$O"
n=0
for cmd in "$@"; do
    n=$((n + 1)); nn=$(printf %02d $n); T=600
    if [[ "$cmd" == T=* ]]; then T=${cmd%% *}; T=${T#T=}; cmd=${cmd#* }; fi
    t0=$(date +%s)
    case "$cmd" in
        TESTS*)    k=${cmd#TESTS}; k=${k#:}
                   if [ -n "$k" ]; then timeout "$T" python -m pytest tests -m gpu -x -q -k "$k" > "$O/$nn.log" 2>&1; else timeout "$T" python -m pytest tests -m gpu -x -q --durations=25 > "$O/$nn.log" 2>&1; fi
                   echo "rc $?" >> "$O/$nn.log" ;;
        BENCH*)    a=${cmd#BENCH}; a=${a#:}
                   timeout "$T" python bench.py $a > "$O/$nn.json" 2> "$O/$nn.err"; echo "rc $?" > "$O/$nn.log"; tail -c 600 "$O/$nn.err" >> "$O/$nn.log"
                   python - "$O/$nn.json" >> "$O/$nn.log" 2>&1 <<'PY'
import json, sys
try:
    j = json.loads([l for l in open(sys.argv[1]).read().splitlines() if l.startswith("{")][-1])
    print("value %.0f  ms %.3f  frac %.4f  stages %s" % (j["value"], j["ms_per_step"], j["roofline"]["frac"], {k: round(v, 3) for k, v in j["stages_ms"].items()}))
    if "msc_drain" in j: print("  msc_drain", {kk: (round(vv, 3) if isinstance(vv, float) else vv) for kk, vv in j["msc_drain"].items() if kk != "what"})
    for k, v in (j.get("extras") or {}).items():
        if isinstance(v, dict):
            print(" ", k, {kk: (round(vv, 3) if isinstance(vv, float) else vv) for kk, vv in v.items() if kk in ("value", "ms_per_step", "parity", "parity_error", "error", "wide_sync_stats", "replayed_batches", "demod_ms", "msc_viterbi_ms", "ms_per_step_all_services_on_the_host")})
except Exception as e:
    print("no bench line:", e)
PY
                   ;;
        PROFILES)  timeout "$T" bash tools/make_profiles.sh > "$O/$nn.log" 2>&1; echo "rc $?" >> "$O/$nn.log" ;;
        *)         timeout "$T" bash -c "$cmd" > "$O/$nn.log" 2>&1; echo "rc $?" >> "$O/$nn.log" ;;
    esac
    echo "== [$nn] ($(( $(date +%s) - t0 )) s) $cmd"; tail -n 6 "$O/$nn.log"
done
