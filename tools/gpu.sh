#!/bin/bash
# The one GPU launcher (through gpurun): `bash tools/gpu.sh <tag> <job> [<job> ...]` runs the jobs in order on the box, each under
# its own timeout, logs into gpurun_out/<tag>/NN_<kind>.log and prints a short tail of each.  Jobs (kind:arguments):
#   suite                      the whole `-m gpu` suite (APEX_RECORD_MEASURED -> gpurun_out/<tag>/measured.jsonl)
#   pytest:<args>              python -m pytest <args> -m gpu
#   smoke                      __graft_entry__.smoke()
#   bench:<bench.py args>      python bench.py <args>           (stdout -> NN_bench.json)
#   py:<script and args>       python <script and args>
#   sh:<command>               bash -c <command>
#   profile:<tag> [cmd]        tools/gpu_profile.sh (rocprofv3 kernel stats + PMC passes + derived CSV)
#   pmc_gemm[:qwen] | pmc_wan | pmc_vae[:flux]    the hash-matched PMC records bench.py reads (tools/gpu_pmc_*.sh -> copy to profiles/)
#   pmc_attn                                      wave-state counters of the attention kernels
# T=<seconds> in front of a job overrides its timeout (default 900):  "T=1800 suite".
# Replaces the one-shot tools/gpu_r0N_*.sh launchers of rounds 3-5 (their measurements live on under profiles/).
set -u
cd "$GRAFT_REPO_ROOT"
TAG=${1:?tag}; shift
OUT=gpurun_out/$TAG
mkdir -p "$OUT"
export ROUND=${ROUND:-r06}
n=0
for job in "$@"; do
    n=$((n + 1))
    T=900
    if [[ "$job" == T=* ]]; then T=${job%% *}; T=${T#T=}; job=${job#* }; fi
    kind=${job%%:*}
    arg=""; [[ "$job" == *:* ]] && arg=${job#*:}
    log=$(printf "%s/%02d_%s" "$OUT" "$n" "$kind")
    t0=$(date +%s)
    case "$kind" in
        suite)   APEX_RECORD_MEASURED=$PWD/$OUT/measured.jsonl timeout "$T" python -m pytest tests/ -m gpu -q -x -p no:cacheprovider > "$log.log" 2>&1 ;;
        pytest)  APEX_RECORD_MEASURED=$PWD/$OUT/measured.jsonl timeout "$T" python -m pytest $arg -m gpu -q -p no:cacheprovider -s > "$log.log" 2>&1 ;;
        smoke)   timeout "$T" python -c "import __graft_entry__ as g; g.smoke()" > "$log.log" 2>&1 ;;
        bench)   timeout "$T" python bench.py $arg > "$log.json" 2> "$log.log" ;;
        py)      timeout "$T" python $arg > "$log.log" 2>&1 ;;
        sh)      timeout "$T" bash -c "$arg" > "$log.log" 2>&1 ;;
        profile) PROF_TIMEOUT=$T bash tools/gpu_profile.sh $arg > "$log.log" 2>&1 ;;
        pmc_gemm) WORKLOAD=${arg:-flux} PROF_TIMEOUT=$T bash tools/gpu_pmc_gemm.sh > "$log.log" 2>&1 ;;
        pmc_attn) PROF_TIMEOUT=$T bash tools/gpu_pmc_attn.sh $arg > "$log.log" 2>&1 ;;
        pmc_wan)  PROF_TIMEOUT=$T bash tools/gpu_pmc_wan.sh > "$log.log" 2>&1 ;;
        pmc_vae)  VAE=${arg:-wan} PROF_TIMEOUT=$T bash tools/gpu_pmc_vae.sh > "$log.log" 2>&1 ;;
        *) echo "unknown job kind '$kind'" > "$log.log" ;;
    esac
    rc=$?
    echo "== [$n] $job -> rc $rc ($(( $(date +%s) - t0 )) s)"
    if [ -s "$log.json" ]; then tail -1 "$log.json" | cut -c1-600; fi
    grep -v "amdgpu.ids" "$log.log" 2>/dev/null | tail -${TAIL:-6}
done
