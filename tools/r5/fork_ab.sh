#!/bin/bash
# Runs ON THE GPU BOX: where the fixed models' branch forks off (SCPR_FORK_AT 0 / 1 / 2): headline and 4K key frames
one() { python3 bench.py "$@" --no-cpu --no-host-boundary 2>/dev/null | python3 -c "
import json,sys,os;d=json.loads(sys.stdin.read());s=d['config']['stage_ms_per_step'];print('fork_at', os.environ.get('SCPR_FORK_AT'), d['value'],d['config']['enc_MPix_s_rank0'],{k:s[k] for k in ('sort','fixed_chain','colour_chain','rans')})"; }
for f in 0 1 2 0 1 2; do export SCPR_FORK_AT=$f; one --no-others --steps 5 --warmup 1; done
for f in 0 1 2; do export SCPR_FORK_AT=$f; one --width 3840 --height 2160 --frames 150 --steps 3 --warmup 1; done
