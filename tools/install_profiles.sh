#!/usr/bin/env bash
# tools/install_profiles.sh <tag> [round] — copy the summaries tools/final_profile.sh <tag> left under gpurun_out/ into profiles/<round>_*
set -e
tag=${1:?tag}; rnd=${2:-r04}
cd "$(dirname "$0")/.."
for id in sponza sponza-dev-wide s10m s10m-dev-wide; do
  for f in hbm_traffic pmc_wf_extend pmc_wf_shade pmc_wf_extend_packet bench; do
    [ -f gpurun_out/${tag}_${f}_${id}.json ] && cp gpurun_out/${tag}_${f}_${id}.json profiles/${rnd}_${f}_${id}.json
  done
  [ -f gpurun_out/${tag}_kernel_stats_${id}.csv ] && cp gpurun_out/${tag}_kernel_stats_${id}.csv profiles/${rnd}_kernel_stats_${id}.csv
done
[ -f gpurun_out/${tag}_hbm_stream.txt ] && cp gpurun_out/${tag}_hbm_stream.txt profiles/${rnd}_hbm_stream.txt
ls profiles/${rnd}_*
