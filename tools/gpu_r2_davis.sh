#!/bin/bash
cd "$GRAFT_REPO_ROOT" && mkdir -p gpurun_out
export TMPDIR=/tmp
TAG=${1:-r2}
timeout 900 python -m pytest tests/test_exact_f32.py tests/test_emu_labelprop.py tests/test_siamfc.py -m gpu -q -p no:cacheprovider > gpurun_out/${TAG}_pytest_lp.txt 2>&1; tail -4 gpurun_out/${TAG}_pytest_lp.txt
for M in r18 r50; do for P in fp32 bf16; do
  timeout 600 python tools/bench_davis.py --model $M --frames 23 --parity-frames 2 --precision $P > gpurun_out/${TAG}_davis_${M}_${P}.json 2> gpurun_out/${TAG}_davis_${M}_${P}.err
  python - gpurun_out/${TAG}_davis_${M}_${P}.json <<'PY'
import json, sys
d = json.load(open(sys.argv[1]))
print(d['model'], d['precision'], 'ms/frame %.3f' % d['ms_per_frame_end_to_end'], 'backbone %.3f' % d['ms_backbone_per_frame'], 'labelprop(%d keys) %.3f ms' % (d['key_frames'], d['ms_labelprop_kernel_21_key_frames']),
      'TFLOP/s %.1f' % d['labelprop_algorithmic_TFLOPs'], 'frac %.3f' % d['labelprop_frac_of_mfma_peak'], 'mismatch', d['label_mismatch_vs_oracle_same_features'])
PY
done; done
