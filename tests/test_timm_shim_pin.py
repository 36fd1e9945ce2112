"""CPU: third-party pin of oracle/timm_shim.py.  The ViT-B goldens come from the reference's vendored VisionTransformer running over
a restatement of the few timm-1.0.9 layers it imports (PatchEmbed, Mlp, DropPath, pos-embed plumbing: timm's source is not under
/root/reference) — written by the same hand as the mirror under test.  HuggingFace transformers (installed here, unrelated to both)
implements the same DINOv2 ViT: on identical weights the reference-over-shim features must equal Dinov2Model's."""
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_reference_vit_over_the_shim_equals_huggingface_dinov2():
    from oracle.ref_import import reference_available
    if not reference_available():
        pytest.skip("reference tree not present (GPU box)")
    import importlib.util
    if importlib.util.find_spec("transformers") is None:
        pytest.skip("transformers not installed")
    import socket
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))     # the loader's world-1 gloo group: not this process's port
    out = subprocess.run([sys.executable, os.path.join(ROOT, "tests", "helpers", "shim_vs_hf.py"), ROOT], capture_output=True, text=True,
                         timeout=600, env=env)
    assert out.returncode == 0, out.stderr[-1500:]
    line = [l for l in out.stdout.splitlines() if l.startswith("MAXDIFF")][-1].split()
    diff, scale = float(line[1]), float(line[3])
    assert scale > 1.0 and diff <= 1e-5 * scale, (diff, scale)
