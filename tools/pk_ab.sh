#!/usr/bin/env bash
# tools/pk_ab.sh — wf_extend_packet against wf_extend on primary rays: default policy (auto), forced on, forced off
run() { # label, env, args...
  label=$1; shift; envs=$1; shift
  env $envs python bench.py --no-cpu-baseline "$@" > gpurun_out/ab_$label.json 2> gpurun_out/ab_$label.err
  python -c "
import json; j=json.load(open('gpurun_out/ab_$label.json')); print('$label', j['value'], j['ms_per_step'])"
}
run sponza_auto RT_X=1 --steps 5 --warmup 2
run sponza_off RT_WF_PACKET=0 --steps 5 --warmup 2
run s10m_auto RT_X=1 --workload s10m --steps 2 --warmup 1
run s10m_on RT_WF_PACKET=1 --workload s10m --steps 2 --warmup 1
run s10m_off RT_WF_PACKET=0 --workload s10m --steps 2 --warmup 1
for spp in 8 16 32; do
  run sp${spp}_auto RT_X=1 --spp $spp --steps 5 --warmup 2
  run sp${spp}_on RT_WF_PACKET=1 --spp $spp --steps 5 --warmup 2
  run sp${spp}_off RT_WF_PACKET=0 --spp $spp --steps 5 --warmup 2
done
