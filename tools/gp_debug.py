"""Per-parameter error table of the gradient-penalty step against the reference fixtures (development aid)."""
import json, os, random, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import test_scalogram_gpu as T
from cpc_audio_amd.audio_dataset import TensorAudioDataset
from cpc_audio_amd.contrastive_estimation_training import ContrastiveEstimationTrainer
gd = os.path.join(ROOT, "tests", "golden")
for fixture in sys.argv[1:] or ["scalogram_model_gp"]:
    g = T._load(gd, fixture + ".npz"); meta = json.load(open(os.path.join(gd, fixture + ".json")))
    data = torch.from_numpy(g["data"])
    for run in meta["runs"]:
        if run.get("gp") is None or run["steps"] != 1:
            continue
        pre, model = T._build_scalogram_model(g, meta, "fp32")
        logger = T._Logger()
        tr = ContrastiveEstimationTrainer(model=model, dataset=TensorAudioDataset(data, device=T.DEV), logger=logger, device=T.DEV,
                                          regularization=run["reg"], score_over_all_timesteps=run["all_timesteps"],
                                          score_function=T.SCORE[run["score"]], prediction_steps=meta["K"], ar_size=meta["H"], preprocessing=pre,
                                          wasserstein_gradient_penalty=True, gradient_penalty_factor=run["gp"])
        tr.verbose = False
        random.seed(run["python_seed"])
        tr.train(batch_size=meta["B"], epochs=10, lr=run["lr"], num_workers=0, max_steps=1)
        print(fixture, run["tag"], "loss", logger.loss_meter.values, "ref", run["loss"])
        scale = max(float(np.abs(g[k]).max()) for k in g if k.startswith(run["tag"] + "/grad/"))
        for k in [k for k in g if k.startswith(run["tag"] + "/grad/")]:
            name = k.split("/grad/")[1]
            got = dict(model.named_parameters())[name].grad.double().cpu()
            ref = torch.from_numpy(g[k]).double()
            l2 = ((got - ref).norm() / (ref.norm() + 1e-30)).item()
            print(f"  {name:60s} l2 {l2:9.2e}  |ref|max {ref.abs().max().item():9.2e} ({ref.abs().max().item() / scale:8.1e} of largest)  |got|max {got.abs().max().item():9.2e}")
