#!/usr/bin/env bash
# tools/final_profile.sh <tag> — everything the round's profiles/ directory is made of, on the GPU box (about 15 minutes):
# both workloads x {parity, production build} through tools/profile_round.sh (bench line, rocprofv3 --stats, FETCH/WRITE_SIZE passes, SQ/TCP/TCC
# passes, stamped summaries) and the HBM stream microbenchmark. Then: tools/install_profiles.sh <tag> r04 copies the summaries
# into profiles/. Run it AFTER the last change to the device sources: bench.py only quotes profiles whose source hash matches.
tag=${1:-r04}
O=gpurun_out; mkdir -p $O
for wl in sponza s10m; do
  ./tools/profile_round.sh $wl $tag parity reference > $O/${tag}_profile_${wl}_parity.txt 2>&1; tail -2 $O/${tag}_profile_${wl}_parity.txt
  ./tools/profile_round.sh $wl $tag wide device > $O/${tag}_profile_${wl}_wide.txt 2>&1; tail -2 $O/${tag}_profile_${wl}_wide.txt   # the production build: all on the device
done
(cd tools/ubench && /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 hbm_stream.hip -o hbm_stream 2>/dev/null && ./hbm_stream > ../../$O/${tag}_hbm_stream.txt 2>&1); cat $O/${tag}_hbm_stream.txt
