"""CPU: the workloads bench.py times are the reference's own recipes.  Every BASELINE.json config names a yaml under
/root/reference/configs; bench.CONFIGS and bench.build_train_step must carry the same model / loss / optimizer settings
(xqgan_train.py:285-347 builds the model, VQLoss and both AdamW from these keys)."""
import inspect
import os
import sys

import pytest
import yaml

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
REF_CFG = os.path.join(os.environ.get("XQ_REFERENCE_ROOT", "/root/reference"), "configs")


@pytest.mark.parametrize("name", ["VQ-8192", "VQ-4096", "VP2-16384", "MSVR10P2-4096", "MSBR10P2-4096", "RobustTok"])
def test_bench_config_equals_the_reference_yaml(name):
    if not os.path.isdir(REF_CFG):
        pytest.skip("reference tree not present (GPU box)")
    import bench
    y = yaml.safe_load(open(os.path.join(REF_CFG, name + ".yaml")))
    c = bench.CONFIGS[name]
    assert c["V"] == y["codebook_size"] and c["C"] == y["codebook_embed_dim"] and c["P"] == y["product_quant"]
    assert list(c["pns"]) == list(y["v_patch_nums"]) and c["L"] == y["num_latent_tokens"]
    assert c["enc"] == y["enc_type"] == y["dec_type"] == "dinov2"
    assert abs(c["drop"] - float(y.get("codebook_drop", 0.0))) < 1e-12 and bool(c["half_sem"]) == bool(y.get("half_sem", False))
    assert bool(c.get("lfq", False)) == bool(y.get("lfq", False))
    assert c["B"] * 8 == y["global_batch_size"] == 1024                    # per-GPU batch of the 8-GPU recipe (xqgan_train.py:241)
    assert y["vq_model"] == "VQ-16" and y["semantic_guide"] == "dinov2" and y["abs_pos_embed"] is True
    assert y["encoder_model"] == y["decoder_model"] == "vit_base_patch14_dinov2.lvd142m"
    if name == "RobustTok":                                                # latent perturbation schedule start (configs/RobustTok.yaml:38-42)
        assert c["alpha"] == float(y["alpha"]) and abs(c["beta_lp"] - float(y["beta"])) < 1e-12 and c["delta"] == int(y["delta"])
    if name == "MSBR10P2-4096":     # not a BASELINE.json config (LFQ extra): its yaml trains with weight_decay 5e-5, the bench keeps the
        return                      # BASELINE recipes' 0.0 — only its geometry is checked
    # what build_train_step hard-codes from the yaml / argparse defaults
    src = inspect.getsource(bench.build_train_step)
    assert float(y["lr"]) == 3e-5 and "3e-5 * gbs / 128" in src
    assert float(y["weight_decay"]) == 0.0 and "weight_decay=0.0," in src
    assert float(y["disc_weight_decay"]) == 0.0005 and "weight_decay=0.0005" in src
    assert float(y["lecam_loss_weight"]) == 0.001 and "lecam_loss_weight=0.001" in src
    assert y["disc_type"] == "dinodisc" and y["disc_adaptive_weight"] is True and "disc_adaptive_weight=True" in src
    assert float(y["sem_loss_weight"]) == 0.1 and int(y["start_drop"]) == 3 and "start_drop=3" in src and "sem_loss_weight=0.1" in src
