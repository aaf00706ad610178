#!/usr/bin/env bash
# tools/install_profiles.sh <tag> [round] — copy the summaries tools/final_profile.sh <tag> left under gpurun_out/ into profiles/<round>_*
set -e
tag=${1:?tag}; rnd=${2:-r02}
cd "$(dirname "$0")/.."
for w in sponza s10m; do
  for f in hbm_traffic pmc_wf_extend pmc_wf_shade bench; do cp gpurun_out/${tag}_${f}_${w}.json profiles/${rnd}_${f}_${w}.json; done
  [ -f gpurun_out/${tag}_pmc_wf_extend_packet_${w}.json ] && cp gpurun_out/${tag}_pmc_wf_extend_packet_${w}.json profiles/${rnd}_pmc_wf_extend_packet_${w}.json
  cp gpurun_out/${tag}_kernel_stats_${w}.csv profiles/${rnd}_kernel_stats_${w}.csv
done
cp gpurun_out/${tag}_hbm_stream.txt profiles/${rnd}_hbm_stream.txt
echo "profiles/${rnd}_extend_sections.txt and ${rnd}_write_amp.txt are narrated by hand: compare with gpurun_out/${tag}_extend_sections_*.txt and ${tag}_write_amp.txt"
