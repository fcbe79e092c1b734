python ../train/train_2d_smoke.py "$@"
