"""the shipped YAML restates the reference's configs/male2female.yaml (compared against the copy of the
parsed reference config stored as data in the golden fixture)"""
import json
import os

import yaml

from conftest import GOLDEN, ROOT


def test_shipped_yaml_matches_reference_config_values():
    mine = yaml.safe_load(open(os.path.join(ROOT, "configs", "male2female.yaml")))
    ref = json.load(open(os.path.join(GOLDEN, "step_full_64.json")))["config"]
    skip = {"display_size"}   # the fixture shrank display_size for speed
    for k, v in mine.items():
        if k in skip:
            continue
        assert ref[k] == v, (k, ref[k], v)
    for k in ("gan_w", "gan_cw", "focus_loss", "focus_delta", "focus_upper", "focus_lower", "focus_epsilon", "recon_x_w",
              "alpha", "lr", "beta1", "beta2", "weight_decay", "step_size", "gamma", "G_update", "D_update", "gen", "dis"):
        assert k in mine


def test_synthesised_yamls_differ_from_male2female_only_where_survey_8d_says():
    """selfie2anime / glasses_removal (BASELINE configs[2], [3]) are male2female.yaml with the data keys, the image size and the
    build's own compute_dtype changed -- architecture and every loss / optimizer hyper-parameter identical."""
    base = yaml.safe_load(open(os.path.join(ROOT, "configs", "male2female.yaml")))
    allowed = {"selfie2anime": {"batch_size", "data_root", "data_kind", "compute_dtype"},
               "glasses_removal": {"batch_size", "data_root", "data_kind", "new_size", "crop_image_height", "crop_image_width"}}
    for name, keys in allowed.items():
        cfg = yaml.safe_load(open(os.path.join(ROOT, "configs", name + ".yaml")))
        diff = {k for k in set(base) | set(cfg) if base.get(k) != cfg.get(k)}
        assert diff == keys, (name, diff)
    s = yaml.safe_load(open(os.path.join(ROOT, "configs", "selfie2anime.yaml")))
    g = yaml.safe_load(open(os.path.join(ROOT, "configs", "glasses_removal.yaml")))
    assert s["compute_dtype"] == "bf16" and s["batch_size"] == 8 and s["crop_image_height"] == 256
    assert g["batch_size"] == 4 and g["new_size"] == g["crop_image_height"] == g["crop_image_width"] == 512
