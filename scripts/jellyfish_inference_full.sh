python ../inference/inference_2d_jellyfish.py "$@"
