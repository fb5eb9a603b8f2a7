"""The entry points themselves (north star: "keeping ... the main_pretrain_* entry points so it is a drop-in for that path"): the root scripts are
run as the reference's are -- `python main_pretrain_*.py --config <json> --path_output <dir>` -- for two synthetic steps, and what a user of the
reference would look for afterwards is checked: the run directory `<path_output>/_<task>-<datasets>_<timestamp>` (main_pretrain_mlm.py:239-244),
args.json (agent.py:155-162), the checkpoint names ckpt_violet_pretrain_{dataset}_{part}_{ep}.pt (main_pretrain_task_specific.py:282-297; no log.json
from this agent) and the checkpoint's key set (= the model's state_dict, held to the reference's key list in tests/test_host_logic.py)."""
import glob
import json
import os
import subprocess
import sys

import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
pytestmark = pytest.mark.gpu


def _config(tmp_path):
    from tests.helpers import hf_cfg
    cfg = hf_cfg("micro")
    c = dict(type="pretrain", task="pretrain", dataset=["synthetic"], vis_backbone_size="micro", vis_backbone_init="random", kinetics=400,
             txt_backbone=cfg, fusion_encoder=cfg, tokenizer=cfg, txt_backbone_embed_only=True, fusion_encoder_rand_init=True,
             size_img=224, size_frame=5, size_txt=31, size_batch=2, size_epoch=1, n_workers=0, lr=2e-5, decay=1e-3, max_grad_norm=1.0,
             logging_steps=1, temp=0.05, size_part=1, seed=88)
    p = tmp_path / "args_pretrain_synthetic.json"
    p.write_text(json.dumps(c))
    return str(p)


@pytest.mark.parametrize("script, model_cls", [("main_pretrain_mlm.py", "LAVENDER_Pretrain_MLM"), ("main_pretrain_task_specific.py", "LAVENDER_Pretrain")])
def test_entry_point_runs_two_synthetic_steps_and_leaves_the_reference_file_contract(tmp_path, script, model_cls):
    out = tmp_path / "snapshot"
    env = dict(os.environ, LAV_SYNTH_STEPS="2", PYTHONPATH=ROOT + os.pathsep + os.environ.get("PYTHONPATH", ""))
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    p = subprocess.run([sys.executable, os.path.join(ROOT, script), "--config", _config(tmp_path), "--path_output", str(out)],
                       capture_output=True, text=True, timeout=600, env=env, cwd=str(tmp_path))
    assert p.returncode == 0, p.stdout[-2000:] + p.stderr[-4000:]
    runs = glob.glob(str(out / "_pretrain-synthetic_*"))
    assert len(runs) == 1, (runs, os.listdir(out) if out.exists() else None)
    run = runs[0]
    stamp = os.path.basename(run).rsplit("_", 1)[1]
    assert len(stamp) == 14 and stamp.isdigit()                                     # %Y%m%d%H%M%S
    args = json.load(open(os.path.join(run, "args.json")))
    assert args["task"] == "pretrain-synthetic" and args["path_output"] == run and args["size_batch"] == 2 and args["max_iter"] == 2
    files = sorted(os.listdir(run))
    assert "ckpt_violet_pretrain_init_0_0.pt" in files and "ckpt_violet_pretrain_init_0_1.pt" in files, files
    assert "log.json" not in files                                                  # the pre-training agents write checkpoints only
    assert "Ep 1:" in p.stdout and "mtm" in p.stdout
    import lavender_amd as LA
    from tests.helpers import Tok
    from lavender_amd.args import EasyDict
    ref_keys = set(getattr(LA, model_cls)(EasyDict(args), Tok()).state_dict())
    sd0 = torch.load(os.path.join(run, "ckpt_violet_pretrain_init_0_0.pt"), map_location="cpu")
    sd1 = torch.load(os.path.join(run, "ckpt_violet_pretrain_init_0_1.pt"), map_location="cpu")
    assert set(sd0) == set(sd1) == ref_keys
    moved = [k for k in sd0 if sd0[k].is_floating_point() and not torch.equal(sd0[k], sd1[k])]
    assert len(moved) > 100, len(moved)                                             # two optimizer steps happened between the two files
    assert all(torch.isfinite(v).all() for v in sd1.values() if v.is_floating_point())
