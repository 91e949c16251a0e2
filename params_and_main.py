"""Configuration + driver of the tile workflow (reference ``params_and_main.py:21-180``) on the MI355X hot path.

Edit the globals, then run ``python params_and_main.py``.  ``Create_tiles`` is CPU preprocessing (``create_tiles_unet.py``, numpy + unet_amd.tiffio instead of GDAL/rasterio);
``Train`` and ``Predict`` run on the GPU.
"""
from __future__ import annotations

import time

from unet_amd import xresnet18, xresnet34, xresnet34_deep, xresnet50, xresnet101  # noqa: F401

# ----------------------------------------------------------------------------------- switches (params_and_main.py:22-24)
Create_tiles = True
Train = False
Predict = False

# ----------------------------------------------------------------------------------- tiles (params_and_main.py:31-38)
image_path = "PATH"
mask_path = "PATH"             # None when cutting prediction tiles without a mask
base_dir = "PATH"
patch_size = 400
patch_overlap = 0              # 0.2 with split = [1] to cut prediction tiles of a full image
split = [0.8, 0.2]

# ----------------------------------------------------------------------------------- training (params_and_main.py:45-63)
data_path = base_dir
model_path = "PATH"
description = "Beschirmung_geo_Aug_data"
info = "RGB images"
existing_model = None          # or the path of an exported model: transfer learning
BATCH_SIZE = 4                 # the reference's value for a 16 GB P100; 16 fits an MI355X many times over
EPOCHS = 15
LEARNING_RATE = 0.0001
enable_regression = False
visualize_data_example = True  # plots are out of scope on this path (ignored)
export_model_summary = True
CODES = ["NO_Data", "Background", "Beschirmung"]
CLASS_WEIGHTS = "even"         # list, "even" or "weighted"

# ----------------------------------------------------------------------------------- prediction (params_and_main.py:68-76)
predict_path = "PATH"
predict_model = "PATH"         # model_path/description/description.pkl
AOI = "str"
year = "str"
merge = False
regression = False
validation_vision = True       # confusion-matrix figures: out of scope on this path (ignored)

# ----------------------------------------------------------------------------------- extra parameters (params_and_main.py:84-118)
enable_extra_parameters = True
self_attention = True
ENCODER_FACTOR = 10
LR_FINDER = None               # None, "minimum", "steep", "valley", "slide"
VALID_SCENES = ["vali"]
loss_func = None               # None = CrossEntropyLossFlat(axis=1) (the reference's default object) | FocalLossFlat(gamma=2, axis=1) (from unet_amd.learner);
                               # regression: MSELossFlat(axis=1), L1LossFlat
monitor = "valid_loss"         # 'dice_multi', 'r2_score', 'train_loss', 'valid_loss'
all_classes = False
specific_class = None
large_file = False
max_empty = 0.2
class_zero = False
ARCHITECTURE = xresnet34
transforms = True
split_idx = 0
n_transform_imgs = 1
# None = the reference's default pipeline HorizontalFlip(p=0.5) + VerticalFlip(p=0.5); or, with `from unet_amd import augment as A`:
# A.Compose([A.HorizontalFlip(p=0.5), A.VerticalFlip(p=0.5),
#            A.RandomBrightnessContrast(brightness_limit=(-0.1, 0.1), contrast_limit=(-0.1, 0.1), p=0.5), A.CoarseDropout(p=0.5)])
aug_pipe = None


def main():
    global large_file, specific_class, all_classes, transforms, VALID_SCENES, self_attention, monitor, loss_func, LR_FINDER
    global ENCODER_FACTOR, ARCHITECTURE, enable_regression, max_empty
    t0 = time.time()
    if enable_extra_parameters:          # params_and_main.py:129-145
        import warnings
        warnings.warn("Extra parameters are enabled. Code may behave in unexpected ways. "
                      "Please disable unless experienced with the code.")
    else:
        ENCODER_FACTOR, LR_FINDER, VALID_SCENES, loss_func, monitor = 10, None, ["vali"], None, None
        all_classes, specific_class, enable_regression, large_file, max_empty = False, None, False, False, 0.9
        ARCHITECTURE, self_attention = xresnet34, False
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
