python ../inference/inference_2d_jellyfish.py --only_vis_pressure "$@"
