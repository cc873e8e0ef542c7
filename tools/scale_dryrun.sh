#!/bin/bash
# 8-GPU readiness on ONE GPU: `bench.py --gpus 8` end to end -- rank launch, tile assignment, per-rank rendering, the N > 1 branch of
# skh_gather_tiles below the C ABI, the root's de-tiling scatter, the max-over-ranks timing -- with the command lines the driver's scaling run uses, at
# FULL resolution (C3 1920x1080 depth 4, C4 3840x2160 depth 6; --spp 4 keeps eight contexts' queues and eight BVH builds on one GPU short), through
# the librccl TEST DOUBLE (tests/cpp/rccl_double.cpp bound by SKH_RCCL_LIB: same call sequence, buffers and sizes as RCCL; shared-memory transport).
# Asserts rccl_nranks == 8 and the gathered image's CRC == the 1-rank image's; prints per_rank.rays_imbalance_max_over_mean and the per-rank times.
# usage (GPU box): bash tools/scale_dryrun.sh [spp]        No scaling number comes out of this: eight ranks share one GPU.
set -e
cd "$(dirname "$0")/.."
SPP=${1:-4}
OUT=${OUT:-gpurun_out/scale_dryrun}
mkdir -p "$OUT"
g++ -std=c++17 -O2 -fPIC -shared -D__HIP_PLATFORM_AMD__ -I/opt/rocm/include -o "$OUT/librccl_double.so" tests/cpp/rccl_double.cpp -L/opt/rocm/lib -lamdhip64 -lrt
export SKH_BENCH_CHECKSUM=1 SKH_DIST_BACKEND=gloo MASTER_ADDR=127.0.0.1 HSA_ENABLE_IPC_MODE_LEGACY=0
COMMON="--steps 1 --warmup 1 --spp $SPP --no-pmc --no-drop-in --no-extra --no-cpu-baseline"
rc=0
for cfg in "C3 --width 1920 --height 1080 --depth 4" "C4 --width 3840 --height 2160 --depth 6"; do
  name=${cfg%% *}; args=${cfg#* }
  python bench.py --gpus 1 $args $COMMON > "$OUT/$name.1.json" 2> "$OUT/$name.1.err"
  SKH_GATHER=rccl SKH_RCCL_LIB="$PWD/$OUT/librccl_double.so" MASTER_PORT=$((29600 + RANDOM % 200)) python bench.py --gpus 8 $args $COMMON > "$OUT/$name.8.json" 2> "$OUT/$name.8.err" || { echo "$name: the 8-rank run failed"; tail -5 "$OUT/$name.8.err"; rc=1; continue; }
  python - "$OUT/$name.1.json" "$OUT/$name.8.json" "$name" <<'PY' || rc=1
import json, sys
one = json.loads([l for l in open(sys.argv[1]) if l.startswith("{")][-1]); many = json.loads([l for l in open(sys.argv[2]) if l.startswith("{")][-1])
name = sys.argv[3]
ok = many["n_gpus"] == 8 and many["rccl_nranks"] == 8 and many["image_crc32"] == one["image_crc32"] and many["config"]["rays_per_frame"] == one["config"]["rays_per_frame"] \
    and many["gather"].startswith("skh_gather_tiles") and "gather_error" not in many
pr = many["per_rank"]
print("%s %s: rccl_nranks %d, gather '%s', image CRC %08x (1 rank: %08x), rays/frame %d; per-rank tiles %s, rays imbalance max/mean %.4f, per-rank ms (8 ranks SHARING one GPU: not a timing) min %.1f max %.1f"
      % (name, "OK" if ok else "FAILED", many["rccl_nranks"], many["gather"], many["image_crc32"], one["image_crc32"], many["config"]["rays_per_frame"], pr["tiles"],
         pr["rays_imbalance_max_over_mean"], pr["ms_per_step"]["min"], pr["ms_per_step"]["max"]))
sys.exit(0 if ok else 1)
PY
done
exit $rc
