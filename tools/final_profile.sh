#!/usr/bin/env bash
# tools/final_profile.sh <tag> — everything the round's profiles/ directory is made of, on the GPU box (about 8 minutes):
# both workloads through tools/profile_round.sh, the section census of wf_extend (stamps-only and counter builds), the
# write-amplification probe, the wf_shade section census and the HBM stream microbenchmark. Needs the development variants:
#   ./tools/variants.sh "diag:-DRT_DIAG" "cyc:-DRT_DIAG_CYCLES" "sdiag:-DRT_DIAG_SHADE"
tag=${1:-r02}
O=gpurun_out; mkdir -p $O
./tools/profile_round.sh sponza $tag > $O/${tag}_profile_sponza.txt 2>&1; tail -3 $O/${tag}_profile_sponza.txt
./tools/profile_round.sh s10m $tag > $O/${tag}_profile_s10m.txt 2>&1; tail -3 $O/${tag}_profile_s10m.txt
RT_DIAG_VARIANT=cyc RT_DIAG_NOCOUNTERS=1 python tools/diag_wf.py 16 > $O/${tag}_extend_sections_cycles.txt 2>&1; tail -3 $O/${tag}_extend_sections_cycles.txt
RT_DIAG_VARIANT=diag python tools/diag_wf.py 8 > $O/${tag}_extend_sections_counts.txt 2>&1; tail -4 $O/${tag}_extend_sections_counts.txt
RT_DIAG_VARIANT=sdiag python tools/diag_shade.py 16 > $O/${tag}_shade_sections.txt 2>&1; tail -10 $O/${tag}_shade_sections.txt
./tools/write_amp_probe.sh ${tag}_wa > $O/${tag}_write_amp.txt 2>&1; tail -3 $O/${tag}_write_amp.txt
(cd tools/ubench && /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 hbm_stream.hip -o hbm_stream 2>/dev/null && ./hbm_stream > ../../$O/${tag}_hbm_stream.txt 2>&1); cat $O/${tag}_hbm_stream.txt
