#!/bin/bash
# Runs ON THE GPU BOX: the fixed models' branch on a stream of higher (-1) / the same (0) / lower (1) priority: headline and 4K key frames
one() { python3 bench.py "$@" --no-cpu --no-host-boundary --no-n8 2>/dev/null | python3 -c "
import json,sys,os;d=json.loads(sys.stdin.read());s=d['config']['stage_ms_per_step'];print('s2prio', os.environ.get('SCPR_S2_PRIORITY'), d['value'],d['config']['enc_MPix_s_rank0'],{k:s[k] for k in ('sort','fixed_chain','colour_chain','rans')})"; }
for f in -1 0 1 -1 0 1; do export SCPR_S2_PRIORITY=$f; one --no-others --steps 5 --warmup 1; done
for f in -1 0 1; do export SCPR_S2_PRIORITY=$f; one --no-others --width 3840 --height 2160 --frames 150 --steps 3 --warmup 1; done
