#!/bin/bash
# Per-phase clock budget of the online LDS kernel (k_online4) on BASELINE config 3's stage: builds tools/online_budget.hip without
# stamps, with level-1 and with level-2 stamps (see lws_online.hip: LAB / LAB2), runs the three on the GPU and writes one JSON.
#   here (no GPU):   bash tools/online_budget.sh build
#   on the GPU box:  bash tools/online_budget.sh run [out.json]        (default gpurun_out/online_phase_budget.json)
cd "$(dirname "$0")/.."
FLAGS="-O3 -std=c++17 --offload-arch=gfx950 -ffp-contract=off -mllvm -amdgpu-sched-strategy=max-ilp -I include -I lws_amd/csrc"
if [ "$1" = "build" ] || [ ! -x tools/online_budget_l1 ]; then
  hipcc $FLAGS tools/online_budget.hip -o tools/online_budget_l0 || exit 1
  hipcc $FLAGS -DLWS_LAB=1 tools/online_budget.hip -o tools/online_budget_l1 || exit 1
  hipcc $FLAGS -DLWS_LAB=2 tools/online_budget.hip -o tools/online_budget_l2 || exit 1
fi
[ "$1" = "build" ] && exit 0
OUT=${2:-gpurun_out/online_phase_budget.json}
mkdir -p "$(dirname "$OUT")"
for l in 0 1 2; do timeout 300 ./tools/online_budget_l$l 256 500 513 3 10 3 > /tmp/online_budget_l$l.json || exit 1; done
python3 - "$OUT" <<'PY'
import json, sys
l0, l1, l2 = (json.load(open("/tmp/online_budget_l%d.json" % i)) for i in range(3))
steps = l1["steps"]
ROLE = {"hw3": "projection wave", "hw7": "centre-frame tap wave (shares the projection wave's SIMD)", "hw0": "tap wave, frame rho-1 (KIND 1)"}
def tick_rate(d):   # counter ticks per ms: the projection wave's phases add up to the whole step loop
    pw = d["waves"]["hw3"]
    return sum(pw) * d["steps"] / d["kernel_ms"]
out = {
 "_what": "s_memtime stamps summed per phase by every wave of workgroup 0 of k_online4<4,5,...,ODD> (config 3's online stage: 256 x 500 x 513, "
          "look-ahead 3, 10 iterations), divided by the number of barrier-delimited steps; tools/online_budget.sh.  A stamp drains the wave's LDS "
          "operations (s_memtime returns through lgkmcnt): level 1 stamps sit where the wave waits for everything anyway, level 2 adds one in front "
          "of every barrier (the projection wave's counted wait becomes a full one).  phases: projection wave [tap waves' sums in registers | own "
          "terms + tree + two re-projections | (level 2) stores landed and next operands fetched | rest of the step up to the barrier's release]; "
          "tap waves [late cells in registers | sums formed and stored | (level 2) early cells of the next pair in, ready for the barrier | barrier]",
 "steps": steps,
 "kernel_ms": {"no_stamps": l0["kernel_ms"], "level1": l1["kernel_ms"], "level2": l2["kernel_ms"]},
 "ticks_per_step": {"level1": sum(l1["waves"]["hw3"]), "level2": sum(l2["waves"]["hw3"])},
 "ticks_per_ms": tick_rate(l1),
 "ns_per_step_no_stamps": 1e6 * l0["kernel_ms"] / steps,
 "level1_ticks_per_step_by_wave": l1["waves"], "level2_ticks_per_step_by_wave": l2["waves"], "roles": ROLE,
}
json.dump(out, open(sys.argv[1], "w"), indent=1)
print(json.dumps(out, indent=1))
PY
