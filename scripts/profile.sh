#!/bin/bash
# Profile collection of a round, run ON THE GPU BOX from the repo root (one gpurun call per part: PART=a|b|c|d|e|f; g: the 50 M frame's lines alone):
#   ROUND=r06 PART=a bash scripts/profile.sh
# Everything lands under gpurun_out/$ROUND/; `python scripts/collect_profiles.py $ROUND` turns it into profiles/${ROUND}_*.
# (One parametrised pair since round 5; the per-round copies of rounds 1-4 are in the history: git log -- scripts/.)
# rocprofv3 runs: kernel trace + stats in one run, every PMC counter set in a run of its own (the pool refuses
# PMC together with other trace domains); the program after `--` is python3 itself.
set -u
cd /tmp && export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT"
ROUND=${ROUND:-r06}
OUT=gpurun_out/$ROUND
mkdir -p $OUT
HEAD="--steps 20 --warmup 5 --no-cpu-baseline --no-extras"
SHORT="--steps 5 --warmup 8 --no-cpu-baseline --no-extras"   # (8: the tile order has settled, GSR_FLAG_NO_TILE_HISTORY)
PART=${PART:-a}

trace() {   # <name> <bench args...>
  local name=$1; shift
  rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace_$name -o run -- python3 bench.py "$@" > $OUT/trace_$name.log 2>&1
}
pmc() {     # <name> <counters> <bench args...>
  local name=$1; local set=$2; shift 2
  timeout -k 10 300 rocprofv3 --pmc $set --kernel-trace --output-format csv -d $OUT/pmc_$name -o run -- python3 bench.py "$@" > $OUT/pmc_$name.log 2>&1
  local f=$(find $OUT/pmc_$name -name '*counter_collection.csv' | head -1)
  if [ -n "$f" ]; then python3 scripts/pmc_summary.py "$f" > $OUT/pmc_$name.txt; else echo "no counter file" > $OUT/pmc_$name.txt; tail -5 $OUT/pmc_$name.log >> $OUT/pmc_$name.txt; fi
}
SQ="SQ_INSTS_VALU SQ_INSTS_SALU SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_BUSY_CYCLES GRBM_GUI_ACTIVE"
LDS="SQ_INSTS_LDS SQ_INSTS_VMEM SQ_WAIT_INST_ANY SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA"
# the blend's instruction classes (bench.py: blend_issue_fractions) and what its waves wait for
CLS="SQ_INSTS_VALU_ADD_F32 SQ_INSTS_VALU_MUL_F32 SQ_INSTS_VALU_FMA_F32 SQ_INSTS_VALU_TRANS_F32 SQ_INSTS_VALU_ADD_F64 SQ_INSTS_VALU_MUL_F64 SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_CVT"
CLS2="SQ_INSTS_VALU_INT32 SQ_INSTS_VALU_INT64 SQ_INSTS_VALU SQ_INSTS_SALU SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_WAIT_INST_LDS SQ_WAVE_CYCLES"

if [ "$PART" = "a" ]; then
  # 1. the headline command, with everything it prints (this is what the driver's BENCH run will be)
  python3 bench.py --steps 20 --warmup 5 > $OUT/bench_default.json 2> $OUT/bench_default.err
  # 2. kernel traces
  trace head $HEAD
  trace head_precomp $HEAD --colors-precomp
  trace outside $SHORT --pose 0,0,-14
  trace far $SHORT --pose 0,0,-30
  trace bound $SHORT --opacity-scale 0.1
  trace 4k $SHORT --width 3840 --height 2160
  # 3. PMC passes, headline frame (and the colorsPrecomp route: preprocess traffic)
  pmc head_sq "$SQ" $SHORT
  pmc head_fetch "FETCH_SIZE" $SHORT
  pmc head_write "WRITE_SIZE" $SHORT
  pmc head_lds "$LDS" $SHORT
  pmc precomp_fetch "FETCH_SIZE" $SHORT --colors-precomp
  pmc precomp_write "WRITE_SIZE" $SHORT --colors-precomp
  pmc head_cls "$CLS" $SHORT
  pmc head_cls2 "$CLS2" $SHORT
fi
if [ "$PART" = "b" ]; then
  # 4. PMC passes on the blend-bound frames (the blend kernel is what is read from these) and the 50 M frame's preprocess
  pmc bound_sq "$SQ" $SHORT --opacity-scale 0.1
  pmc outside_sq "$SQ" $SHORT --pose 0,0,-14
  pmc far_sq "$SQ" $SHORT --pose 0,0,-30
  pmc far_cls "$CLS" $SHORT --pose 0,0,-30
  pmc far_cls2 "$CLS2" $SHORT --pose 0,0,-30
  pmc far_onewave_sq "$SQ" $SHORT --pose 0,0,-30 --no-deep-tiles
  pmc far_onewave_cls "$CLS" $SHORT --pose 0,0,-30 --no-deep-tiles
  pmc far_onewave_cls2 "$CLS2" $SHORT --pose 0,0,-30 --no-deep-tiles
  pmc bound_cls "$CLS" $SHORT --opacity-scale 0.1
  pmc bound_cls2 "$CLS2" $SHORT --opacity-scale 0.1
  pmc outside_cls "$CLS" $SHORT --pose 0,0,-14
  pmc outside_cls2 "$CLS2" $SHORT --pose 0,0,-14
  trace stress50M $SHORT --scene stress --splats 50000000
  trace stress50M_precomp $SHORT --scene stress --splats 50000000 --colors-precomp
  pmc stress_precomp_fetch "FETCH_SIZE" $SHORT --scene stress --splats 50000000 --colors-precomp
  pmc stress_precomp_write "WRITE_SIZE" $SHORT --scene stress --splats 50000000 --colors-precomp
  pmc stress_fetch "FETCH_SIZE" $SHORT --scene stress --splats 50000000
  pmc stress_write "WRITE_SIZE" $SHORT --scene stress --splats 50000000
fi
if [ "$PART" = "c" ]; then
  # 5. the other configurations (bench lines only)
  python3 bench.py --steps 20 --warmup 8 --no-cpu-baseline --no-extras --backward > $OUT/bench_backward.json 2>/dev/null
  python3 bench.py --steps 10 --warmup 8 --no-cpu-baseline --no-extras --backward --pose 0,0,-14 > $OUT/bench_backward_outside.json 2>/dev/null
  python3 bench.py --steps 20 --warmup 8 --no-cpu-baseline --no-extras --backward --no-sorted-lists > $OUT/bench_backward_nolists.json 2>/dev/null
  python3 bench.py --steps 10 --warmup 8 --no-cpu-baseline --no-extras --width 3840 --height 2160 > $OUT/bench_4k.json 2>/dev/null
  python3 bench.py --steps 10 --warmup 8 --no-cpu-baseline --no-extras --scene stress --splats 50000000 > $OUT/bench_stress50M.json 2>/dev/null
  python3 bench.py --steps 10 --warmup 8 --no-cpu-baseline --no-extras --scene stress --splats 50000000 --colors-precomp > $OUT/bench_stress50M_precomp.json 2>/dev/null
  python3 bench.py --steps 10 --warmup 8 --no-cpu-baseline --no-extras --scene stress --splats 50000000 --semantics inria --sh-degree 3 > $OUT/bench_stress50M_inria_sh3.json 2>/dev/null
  GSR_FORCE_DIST=1 python3 bench.py --steps 10 --warmup 8 --no-cpu-baseline > $OUT/bench_forced_dist_1rank.json 2>/dev/null
  rm -f gpurun_out/band_projection.json
  python3 scripts/band_timings.py 1920 1080 > $OUT/band_timings.txt 2>/dev/null
  python3 scripts/band_timings.py 3840 2160 > $OUT/band_timings_4k.txt 2>/dev/null
  cp gpurun_out/band_projection.json $OUT/band_projection.json
fi
if [ "$PART" = "d" ]; then
  # 6. parity report (both oracle builds, config 5, blend times) and the backward soak
  timeout -k 10 900 python3 tests/fullsize_parity_report.py --timing > $OUT/parity.txt 2>$OUT/parity.err
  timeout -k 10 600 python3 scripts/soak_r02.py 100 10 > $OUT/soak.txt 2>&1
fi
if [ "$PART" = "e" ]; then
  # 7. what the tile history makes of a camera path and of unrelated views; the switch points on a trained-like scene
  python3 scripts/history_similarity.py > $OUT/history_similarity.txt 2>/dev/null
  for a in "trained_like 1000000" "trained_like 5834784" "garden_like 1000000" "garden_like 5834784"; do
    set -- $a
    python3 scripts/thresholds_check.py $1 $2 60 2>/dev/null > $OUT/thresholds_$1_$2.txt
  done
  python3 scripts/clock_ramp.py 2>/dev/null > $OUT/clock_ramp.txt
fi
if [ "$PART" = "f" ]; then
  # 8. parity soaks against the oracle (every third pose from far out: deep tiles), the micro-benchmarks, the deep tiles' table, the PLY path
  timeout -k 10 500 python3 scripts/soak_parity.py 60 1000000 1280 720 trained_like > $OUT/soak_trained_like.txt 2>&1
  ./scripts/micro/gather_dc > $OUT/micro_gather_dc.txt 2>&1
  ./scripts/micro/scatter_records > $OUT/micro_scatter_records.txt 2>&1
  ./scripts/micro/xcd_placement > $OUT/micro_xcd_placement.txt 2>&1
  ./scripts/micro/event_gap > $OUT/micro_event_gap.txt 2>&1
  ./scripts/micro/valu_issue > $OUT/micro_valu_issue.txt 2>&1
  python3 scripts/deep_tiles_table.py 2>/dev/null > $OUT/deep_tiles.txt
  python3 scripts/path_stages.py 2>/dev/null > $OUT/path_stages.txt
  timeout -k 10 400 python3 scripts/ply_path.py 2>/dev/null > $OUT/ply_path.txt
  timeout -k 10 500 python3 scripts/soak_parity.py 60 1000000 1280 720 garden_like > $OUT/soak_garden_like.txt 2>&1
fi
if [ "$PART" = "g" ]; then
  # 9. the 50 M frame's lines alone (after a change that touches only the route beyond 16 M Gaussians)
  trace stress50M $SHORT --scene stress --splats 50000000
  trace stress50M_precomp $SHORT --scene stress --splats 50000000 --colors-precomp
  pmc stress_precomp_fetch "FETCH_SIZE" $SHORT --scene stress --splats 50000000 --colors-precomp
  pmc stress_precomp_write "WRITE_SIZE" $SHORT --scene stress --splats 50000000 --colors-precomp
  pmc stress_fetch "FETCH_SIZE" $SHORT --scene stress --splats 50000000
  pmc stress_write "WRITE_SIZE" $SHORT --scene stress --splats 50000000
  python3 bench.py --steps 10 --warmup 8 --no-cpu-baseline --no-extras --scene stress --splats 50000000 > $OUT/bench_stress50M.json 2>/dev/null
  python3 bench.py --steps 10 --warmup 8 --no-cpu-baseline --no-extras --scene stress --splats 50000000 --colors-precomp > $OUT/bench_stress50M_precomp.json 2>/dev/null
  python3 bench.py --steps 10 --warmup 8 --no-cpu-baseline --no-extras --scene stress --splats 50000000 --semantics inria --sh-degree 3 > $OUT/bench_stress50M_inria_sh3.json 2>/dev/null
fi
ls -la $OUT | head -80
