python ../train/train_2d_smoke.py \
--is_w_model "$@"
