"""Configuration + driver of the tile workflow (reference ``params_and_main.py:21-180``) on the MI355X hot path.

Edit the globals, then run ``python params_and_main.py``.  ``Create_tiles`` is CPU preprocessing (``create_tiles_unet.py``, numpy + unet_amd.tiffio instead of GDAL/rasterio);
``Train`` and ``Predict`` run on the GPU.
"""
from __future__ import annotations

import time

from unet_amd import xresnet18, xresnet34, xresnet50  # noqa: F401

# ----------------------------------------------------------------------------------- switches (params_and_main.py:22-24)
Create_tiles = False
Train = True
Predict = False

# ----------------------------------------------------------------------------------- tiles (params_and_main.py:27-47)
image_path = "PATH"
mask_path = "PATH"
base_dir = "PATH"
patch_size = 400
patch_overlap = 0
split = [0.8, 0.2]
class_zero = False
max_empty = 0.9

# ----------------------------------------------------------------------------------- training (params_and_main.py:49-118)
data_path = "PATH"
model_path = "PATH"
description = "unet_run"
info = ""
BATCH_SIZE = 16
EPOCHS = 30
LEARNING_RATE = 1e-4
enable_regression = False
visualize_data_example = False
export_model_summary = True
enable_extra_parameters = False
self_attention = False
ENCODER_FACTOR = 10
LR_FINDER = None
VALID_SCENES = ["vali"]
loss_func = None
monitor = "dice_multi"
CLASS_WEIGHTS = "even"
ARCHITECTURE = xresnet34
existing_model = None
CODES = ["background", "class1", "class2", "class3", "class4"]
transforms = False
n_transform_imgs = 1
aug_pipe = None
split_idx = None

# ----------------------------------------------------------------------------------- prediction
predict_path = "PATH"
predict_model = "PATH"
AOI = None
year = None
merge = False
all_classes = False
specific_class = None
large_file = False
validation_vision = False
regression = False


def main():
    global self_attention, ENCODER_FACTOR, LR_FINDER, loss_func, monitor, ARCHITECTURE, transforms
    t0 = time.time()
    if not enable_extra_parameters:      # params_and_main.py:130-146: reset the "expert" knobs
        self_attention, ENCODER_FACTOR, LR_FINDER, loss_func, monitor = False, 10, None, None, "dice_multi"
        ARCHITECTURE, transforms = xresnet34, False
    if Create_tiles:
        from create_tiles_unet import split_raster
        split_raster(path_to_raster=image_path, path_to_mask=mask_path, base_dir=base_dir, patch_size=patch_size,
                     patch_overlap=patch_overlap, split=split, max_empty=max_empty, class_zero=class_zero)
    if Train:
        from train import train_func
        train_func(data_path, existing_model, model_path, description, BATCH_SIZE, visualize_data_example, enable_regression,
                   CLASS_WEIGHTS, ARCHITECTURE, EPOCHS, LEARNING_RATE, ENCODER_FACTOR, LR_FINDER, loss_func, monitor, self_attention,
                   VALID_SCENES, CODES, transforms, split_idx, export_model_summary, aug_pipe, n_transform_imgs, info, class_zero)
    if Predict:
        from predict import save_predictions
        save_predictions(predict_model, predict_path, regression, merge, all_classes, specific_class, large_file, AOI, year,
                         validation_vision, class_zero)
    print(f"Operation completed in {(time.time() - t0) / 60:.2f} minutes.")


if __name__ == "__main__":
    main()
