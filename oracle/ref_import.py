"""TEST INFRASTRUCTURE ONLY — import shim for the *unmodified* reference modules.

Only usable in the build container where /root/reference exists (it does not travel to the
GPU box).  Used by oracle/make_golden.py to generate tests/golden/*.npz and by the
"pin the oracle" tests (skipped when the reference is absent).  Nothing in the product
package (imagefolder_amd/) may import this file.

What the shim does (documented in SURVEY.md §8c / Appendix D):
  * puts reference/tokenizer/tokenizer_image first on sys.path (quant.py:10 does `import dist`)
    and /root/reference second, ahead of site-packages (xqgan_model.py:25 `from datasets import
    Denormalize` must not resolve to HuggingFace `datasets`);
  * stubs the python deps that are absent offline (timm, peft, torchvision, webdataset, wandb);
  * creates a gloo world_size=1 process group (the quantizers call tdist.all_reduce /
    get_world_size unconditionally: xqgan_model.py:775,786; quant.py:104,137).
"""
import os
import sys
import importlib.util
from unittest.mock import MagicMock

REF_ROOT = os.environ.get("XQ_REFERENCE_ROOT", "/root/reference")


def reference_available() -> bool:
    return os.path.isdir(os.path.join(REF_ROOT, "tokenizer", "tokenizer_image"))


_loaded = {}


def load_reference():
    """Returns a dict of the reference symbols on the hot path (imported unmodified)."""
    if _loaded:
        return _loaded
    if not reference_available():
        raise RuntimeError("reference tree not present at %s" % REF_ROOT)
    saved_path = list(sys.path)
    saved_datasets = sys.modules.pop("datasets", None)
    sys.path.insert(0, os.path.join(REF_ROOT, "tokenizer", "tokenizer_image"))
    sys.path.insert(1, REF_ROOT)
    for m in ["peft", "torchvision", "torchvision.datasets", "torchvision.transforms",
              "torchvision.models", "torchvision.utils", "webdataset", "wandb"]:
        if m not in sys.modules:
            sys.modules[m] = MagicMock()
    # timm: functional stand-in for the handful of layers the vendored ViT needs (oracle/timm_shim.py)
    from oracle import timm_shim
    timm_shim.install()
    import torch.distributed as tdist
    if not tdist.is_initialized():
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29511")
        tdist.init_process_group("gloo", rank=0, world_size=1)
    import warnings
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        from tokenizer.tokenizer_image import xqgan_model as ref_model
        from tokenizer.tokenizer_image import quant as ref_quant
        from tokenizer.tokenizer_image import latent_perturbation as ref_lp
        from tokenizer.tokenizer_image import lookup_free_quantize as ref_lfq
    spec = importlib.util.spec_from_file_location(
        "ref_models_quant", os.path.join(REF_ROOT, "models", "quant.py"))
    ref_var_quant = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(ref_var_quant)
    _loaded.update(
        xqgan_model=ref_model, quant=ref_quant, latent_perturbation=ref_lp,
        var_quant=ref_var_quant,
        dinov2=sys.modules["tokenizer.tokenizer_image.dino_enc.dinov2"],
        DINOv2Encoder=ref_model.DINOv2Encoder, DINOv2Decoder=ref_model.DINOv2Decoder,
        VectorQuantizer=ref_model.VectorQuantizer, VectorQuantizer2=ref_quant.VectorQuantizer2,
        add_perturbation=ref_lp.add_perturbation, VQ_models=ref_model.VQ_models,
        Encoder=ref_model.Encoder, Decoder=ref_model.Decoder, VQModel=ref_model.VQModel,
        lookup_free_quantize=ref_lfq, LFQ=ref_lfq.LFQ,
    )
    # keep the reference dirs on sys.path only as long as needed for lazy imports inside the
    # reference; restore ordering so HF `datasets` & co. are reachable again for other code.
    sys.path[:] = saved_path + [p for p in sys.path if p not in saved_path and p.startswith(REF_ROOT)]
    if saved_datasets is not None:
        sys.modules["datasets"] = saved_datasets
    return _loaded


def load_reference_evaluator():
    """The reference's evaluator.py (ADM FID evaluator) imported unmodified with its absent heavy dependencies stubbed
    (tensorflow, requests, tqdm are only touched by the Inception-graph code paths): gives the real `FIDStatistics.frechet_distance`
    (evaluator.py:72-115) and `Evaluator.compute_statistics` (:186-189) to pin imagefolder_amd.rfid against."""
    if "evaluator" in _loaded:
        return _loaded["evaluator"]
    if not reference_available():
        raise RuntimeError("reference tree not present at %s" % REF_ROOT)
    for m in ["tensorflow", "tensorflow.compat", "tensorflow.compat.v1", "requests"]:
        if m not in sys.modules:
            sys.modules[m] = MagicMock()
    spec = importlib.util.spec_from_file_location("ref_evaluator", os.path.join(REF_ROOT, "evaluator.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    _loaded["evaluator"] = mod
    return mod
