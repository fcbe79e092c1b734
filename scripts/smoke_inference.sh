python ../inference/inference_2d_smoke.py "$@"
