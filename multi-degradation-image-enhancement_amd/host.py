"""Boundary caller: the pieces of the reference's driver that sit directly on the hot path's boundary
(SURVEY.md 8b / section 2 rows 3-6), rebuilt so that `run.py -c <config>.json -p test` drives the HIP
engine with the reference's own config files:

  load_config / instantiate      utils/parser.py:28-73   (JSON with // comments, None for missing keys,
                                                           ["module", "Class"] factory)
  ImageFolderPairs / ImageFolder  data/dataset.py:29-112  (pairing modes; PIL decode)
  Model                           models/model.py:25-363, models/base.py:11-55 (test path)
  RunLogger                       utils/logger.py:33-166  (csv / jsonl / summary rows)

Differences that matter on MI355X: the `Normalize(mean 0, std 1) -> ToTensorV2` tail of the configs'
transform list is executed ON THE GPU (uint8 HWC batches cross PCIe pinned and asynchronously, 4x fewer
bytes than fp32; `mdie_u8hwc_to_f32nchw`), post-processing / uint8 conversion / PSNR / SSIM run as HIP
kernels on the output while it is still in HBM.  LPIPS and the VGG loss need downloaded weights and are
skipped with a warning.  `Model.train()` runs the training-mode graph of mdie_amd.train (HIP convolutions
forward/dgrad/wgrad under autograd) with Adam and, under torch.distributed, bucketed gradient all-reduce.
"""
import csv
import importlib
import json
import os
import time
import warnings
from collections import OrderedDict

import numpy as np
from . import launch
import torch
from PIL import Image
from torch.utils.data import DataLoader, Dataset

IMAGE_EXT = (".png", ".jpg", ".jpeg", ".bmp", ".tif", ".tiff", ".webp")


# ---- configuration ------------------------------------------------------------------------------------------------
class Cfg(dict):
    """dict whose missing keys read as None (the reference's NoneDict contract, utils/parser.py:10-12)."""

    def __missing__(self, key):
        return None


def _wrap(node):
    if isinstance(node, dict):
        return Cfg((k, _wrap(v)) for k, v in node.items())
    if isinstance(node, list):
        return [_wrap(v) for v in node]
    return node


def load_config(path, phase):
    with open(path, "r") as f:
        text = "\n".join(line.split("//")[0] for line in f.read().splitlines())
    cfg = json.loads(text, object_pairs_hook=OrderedDict)
    cfg["phase"] = phase
    return _wrap(cfg)


def instantiate(spec, *args, default_module=None, kind="Network", **extra):
    """spec = {"name": ["module.path", "Attr"] | "Attr", "args": {...}}; failures surface as
    NotImplementedError like the reference's factory (utils/parser.py:69-71) but keep their cause."""
    name = spec["name"]
    module_name, attr_name = (name[0], name[1]) if isinstance(name, (list, tuple)) else (default_module, name)
    try:
        attr = getattr(importlib.import_module(module_name), attr_name)
        kwargs = dict(spec.get("args") or {})
        kwargs.update(extra)
        obj = attr(*args, **kwargs)
        try:
            obj.__name__ = type(obj).__name__
        except Exception:
            pass
        return obj
    except Exception as e:  # noqa: BLE001
        raise NotImplementedError(f"{kind} [{attr_name}() from {module_name}] not recognized.") from e


# ---- datasets ----------------------------------------------------------------------------------------------------------
def _images(root):
    return sorted(f for f in os.listdir(root) if f.lower().endswith(IMAGE_EXT) and not f.startswith("."))   # data/dataset.py:19


def _resize_bilinear(x, h, w):
    """uint8 HWC -> uint8 HWC, bilinear with half-pixel centres and NO antialiasing: what albumentations' Resize does
    (cv2.INTER_LINEAR).  PIL's BILINEAR widens its filter when shrinking, which changes test-phase PSNR/SSIM."""
    if x.shape[0] == h and x.shape[1] == w:
        return x
    t = torch.from_numpy(np.array(x, copy=True)).permute(2, 0, 1)[None].float()
    t = torch.nn.functional.interpolate(t, size=(h, w), mode="bilinear", align_corners=False, antialias=False)
    return t[0].permute(1, 2, 0).add_(0.5).clamp_(0, 255).to(torch.uint8).numpy()


def _lut(x, table):
    return np.clip(table, 0, 255).astype(np.uint8)[x]


class _Transform:
    """The op lists of the reference's configs (utils/transforms_factory.py:19-86) on numpy uint8 images.  Geometric ops
    and the photometric ones the shipped configs use (RandomBrightnessContrast, RandomGamma; config/low_light.json:102-103,
    config/low_contrast.json:101) plus GaussNoise follow albumentations' published formulas and, like its
    `additional_targets={"target": "image"}` (:85), draw ONE set of parameters per sample for input and target
    (albumentations itself is not installed here: these are restatements, not pinned against it).  Op names are checked
    at construction.  If the list ends with Normalize(mean 0, std 1) + ToTensorV2 (every shipped config does), that tail
    is left to the GPU and samples stay uint8 HWC."""

    SUPPORTED = ("Resize", "HorizontalFlip", "RandomHorizontalFlip", "VerticalFlip", "RandomVerticalFlip", "RandomRotate90",
                 "RandomBrightnessContrast", "RandomGamma", "GaussNoise", "Normalize", "ToTensorV2", "ToTensor")

    def __init__(self, cfg, device_tail=True):
        self.ops = list((cfg or {}).get("ops", []))
        names = [o["name"] for o in self.ops]
        bad = [n for n in names if n not in self.SUPPORTED]
        if bad:
            raise ValueError(f"transform not supported by the MI355X host path: {', '.join(bad)} (supported: {', '.join(self.SUPPORTED)})")
        self.on_device = False
        if device_tail and len(names) >= 2 and names[-2:] == ["Normalize", "ToTensorV2"]:
            a = self.ops[-2].get("args") or {}
            if all(abs(m) < 1e-12 for m in a.get("mean", [0, 0, 0])) and all(abs(s - 1) < 1e-12 for s in a.get("std", [1, 1, 1])):
                self.ops, self.on_device = self.ops[:-2], True

    def __call__(self, *imgs, rng=None):
        arrs = [np.asarray(im.convert("RGB")) for im in imgs]
        as_float = False
        for op in self.ops:
            name, a = op["name"], (op.get("args") or {})
            fires = rng is not None and rng.random() < a.get("p", 0.5)
            if name == "Resize":
                h, w = (a["size"] if "size" in a else (a["height"], a["width"]))
                arrs = [_resize_bilinear(x, h, w) for x in arrs]
            elif name in ("HorizontalFlip", "RandomHorizontalFlip"):
                if fires:
                    arrs = [x[:, ::-1] for x in arrs]
            elif name in ("VerticalFlip", "RandomVerticalFlip"):
                if fires:
                    arrs = [x[::-1] for x in arrs]
            elif name == "RandomRotate90":
                if fires:
                    k = int(rng.integers(0, 4))
                    arrs = [np.rot90(x, k) for x in arrs]
            elif name == "RandomBrightnessContrast":       # alpha = 1 + U(-c, c), beta = U(-b, b); uint8: x * alpha + beta * 255
                if fires:
                    b_lim, c_lim = a.get("brightness_limit", 0.2), a.get("contrast_limit", 0.2)
                    alpha, beta = 1.0 + rng.uniform(-c_lim, c_lim), rng.uniform(-b_lim, b_lim)
                    table = np.arange(256, dtype=np.float32) * alpha + beta * 255.0
                    arrs = [_lut(x, table) for x in arrs]
            elif name == "RandomGamma":                    # gamma = U(lo, hi) / 100; uint8: (x / 255) ** gamma * 255
                if fires:
                    lo, hi = a.get("gamma_limit", (80, 120))
                    table = np.power(np.arange(256, dtype=np.float32) / 255.0, rng.uniform(lo, hi) / 100.0) * 255.0
                    arrs = [_lut(x, table) for x in arrs]
            elif name == "GaussNoise":                     # additive N(mean, var) with var = U(var_limit), same noise field for the pair
                if fires:
                    lo, hi = a.get("var_limit", (10.0, 50.0))
                    noise = rng.normal(a.get("mean", 0.0), np.sqrt(rng.uniform(lo, hi)), arrs[0].shape)
                    arrs = [np.clip(x.astype(np.float32) + noise, 0, 255).astype(np.uint8) for x in arrs]
            elif name == "Normalize":
                mean, std = np.asarray(a["mean"], np.float32), np.asarray(a["std"], np.float32)
                arrs = [(x.astype(np.float32) / 255.0 - mean) / std for x in arrs]
                as_float = True
            elif name == "ToTensor" and not as_float:
                arrs = [x.astype(np.float32) / 255.0 for x in arrs]
                as_float = True
        out = []
        for x in arrs:
            x = np.array(x, copy=True)       # (flips / rot90 leave negative-stride views)
            out.append(torch.from_numpy(x) if self.on_device else torch.from_numpy(x.astype(np.float32)).permute(2, 0, 1).contiguous())
        return out


def _sample_rng(holder):
    """Augmentation generator of the calling process: DataLoader workers are re-forked every epoch with a fresh
    torch.initial_seed() (base seed + worker id), the main process keeps its own; the torchrun rank is mixed in so that
    ranks, workers and epochs all draw different flips."""
    key = (os.getpid(), torch.initial_seed())
    if holder.get("key") != key:
        holder["key"] = key
        holder["rng"] = np.random.default_rng([torch.initial_seed() % (2 ** 63), int(os.environ.get("RANK", 0))])
    return holder["rng"]


class ImageFolderPairs(Dataset):
    """degraded/clean folders paired by "filename", "stem" or (legacy) "sorted" -- data/dataset.py:29-92."""

    def __init__(self, input_root, target_root, pairing_mode="filename", transform=None, image_size=None):
        a, b = _images(input_root), _images(target_root)
        if pairing_mode == "sorted":
            self.pairs = [(os.path.join(input_root, x), os.path.join(target_root, y)) for x, y in zip(a, b)]
        elif pairing_mode in ("filename", "stem"):
            key = (lambda f: f) if pairing_mode == "filename" else (lambda f: os.path.splitext(f)[0])
            am, bm = {key(f): f for f in a}, {key(f): f for f in b}
            common = sorted(set(am) & set(bm))
            if not common:
                raise RuntimeError(f"No paired files found with pairing_mode='{pairing_mode}'.\n"
                                   f"input_root={input_root}\ntarget_root={target_root}")
            self.pairs = [(os.path.join(input_root, am[k]), os.path.join(target_root, bm[k])) for k in common]
        else:
            raise ValueError(f"Unknown pairing_mode: {pairing_mode}")
        self.tf = _Transform(transform)
        self._rng = {}

    def __len__(self):
        return len(self.pairs)

    def __getitem__(self, i):
        x, t = self.tf(Image.open(self.pairs[i][0]), Image.open(self.pairs[i][1]), rng=_sample_rng(self._rng))
        return x, t


class ImageFolder(Dataset):
    """data/dataset.py:95-112."""

    def __init__(self, input_root, transform=None):
        self.files = [os.path.join(input_root, f) for f in _images(input_root)]
        self.tf = _Transform(transform)

    def __len__(self):
        return len(self.files)

    def __getitem__(self, i):
        return self.tf(Image.open(self.files[i]))[0]


# ---- one process per GPU: `python -m torch.distributed.run --nproc-per-node N run.py -c ... -p train` ------------
def dist_env():
    """(rank, world, local_rank) from torchrun's environment; (0, 1, 0) for a plain `python run.py`."""
    return int(os.environ.get("RANK", 0)), int(os.environ.get("WORLD_SIZE", 1)), int(os.environ.get("LOCAL_RANK", 0))


def cpu_share():
    """CPUs this process can actually use: the smaller of its affinity mask and its cgroup's CPU quota (cgroup v2 `cpu.max`, v1
    `cpu.cfs_quota_us / cpu.cfs_period_us`).  A one-GPU share of an MI355X host sees all 256 CPUs and is granted 16: sizing a thread pool
    by os.cpu_count() there makes every parallel CPU operation crawl (a 6 MB tensor copy took 60 ms)."""
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    try:
        with open("/sys/fs/cgroup/cpu.max") as f:
            quota, period = f.read().split()[:2]
        if quota != "max":
            n = min(n, max(1, -(-int(quota) // int(period))))
    except Exception:
        try:
            with open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us") as f1, open("/sys/fs/cgroup/cpu/cpu.cfs_period_us") as f2:
                q, p = int(f1.read()), int(f2.read())
            if q > 0 and p > 0:
                n = min(n, max(1, -(-q // p)))
        except Exception:
            pass
    return max(1, n)


def cap_cpu_threads():
    """torch's intra-op pool never larger than cpu_share() (only ever lowered); returns the pool size in force"""
    n = cpu_share()
    if torch.get_num_threads() > n:
        torch.set_num_threads(n)
    return torch.get_num_threads()


def _cpulist(text):
    cpus = set()
    for part in text.strip().split(","):
        if part:
            lo, _, hi = part.partition("-")
            cpus.update(range(int(lo), int(hi or lo) + 1))
    return cpus


def bind_to_gpu_numa(device_index=0):
    """Keep this process on the CPUs of the NUMA node its GPU hangs off (MDIE_NUMA_BIND=0 turns it off).
    A two-socket MI355X host gives a process CPUs of both sockets; where its threads happen to run decides on which node pinned host
    buffers are first touched, and a batch that crosses the socket link on its way to PCIe moves at half the rate -- the PCIe-inclusive
    serving rate was bimodal by process, 26 k or 13.7 k images/s on one box (tools/bench_e2e.py, profiles/LEDGER.md (rounds 1-4) section 7).  The node comes from the
    device's PCI address (torch's device properties -> /sys/bus/pci/devices/<bdf>/local_cpulist); the binding is the intersection with
    the CPUs the process is allowed.  Returns the CPU set bound to, or None when nothing was changed (no sysfs entry, no GPU, a single
    node, an empty intersection, or switched off) -- never raises: placement is speed, not correctness."""
    if os.environ.get("MDIE_NUMA_BIND", "1") == "0" or not hasattr(os, "sched_setaffinity"):
        return None
    try:
        p = torch.cuda.get_device_properties(device_index)
        bdf = f"{getattr(p, 'pci_domain_id', 0):04x}:{p.pci_bus_id:02x}:{p.pci_device_id:02x}.0"
        with open(f"/sys/bus/pci/devices/{bdf}/local_cpulist") as f:
            local = _cpulist(f.read())
        allowed = os.sched_getaffinity(0)
        want = local & allowed
        if not want or want == allowed:
            return None
        os.sched_setaffinity(0, want)
        return want
    except Exception:
        return None


def init_distributed(backend=None):
    """Under torchrun: bind this process to its GPU and join the process group (RCCL over xGMI = backend "nccl";
    "gloo" on a CPU-only host, which only the plumbing tests use).  No-op for a single process."""
    import torch.distributed as dist
    rank, world, local = dist_env()
    if world <= 1 or (dist.is_available() and dist.is_initialized()):
        return rank, world, local
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    # (a launcher -- torchrun, mdie_amd.launch.self_launch -- always sets MASTER_PORT, the latter to a port the kernel has just handed out;
    #  29500 is only the last resort of a hand-started rank: it is where every torch job on a shared host looks first)
    os.environ.setdefault("MASTER_PORT", "29500")
    if torch.cuda.is_available():
        torch.cuda.set_device(local)
        bind_to_gpu_numa(local)
        launch.init_or_exit(dist.init_process_group, backend or "nccl", device_id=torch.device("cuda", local))
    else:
        launch.init_or_exit(dist.init_process_group, backend or "gloo")
    global _OWNS_PROCESS_GROUP
    _OWNS_PROCESS_GROUP = True
    return rank, world, local


_OWNS_PROCESS_GROUP = False      # init_distributed() created the default group: run() then also takes it down, in the order below


def shutdown_distributed(captured=None, buckets=None, destroy=True):
    """THE teardown order of a data-parallel process; every path that owns a process group ends with it (host.run, bench.py,
    tools/bench_train.py, tests/ddp_one_rank_gpu.py).

    What RCCL (= NCCL's rules) and torch require, and what round 5 violated: a hipGraph that captured collectives RETAINS resources of
    the communicator (RCCL attaches a user object to the graph -- hipGraphRetainUserObject -- whose destructor, run by the HIP runtime
    on a thread of its own when the graph and its executables are finally released, hands the captured plans back to the communicator),
    and an async Work handle keeps events of the communication stream.  The communicator must therefore outlive every such graph and
    handle.  Round 5's one abort (`Fatal Python error: Aborted`, main thread inside destroy_process_group(), a second thread with no
    Python frame: gpurun_out/r05a/poison.log) came out of a process that had captured RCCL all-reduces into CapturedStep graphs and went
    into the teardown relying on `del cap` alone -- no collection of reference cycles, a synchronize BEFORE the last graphs were dropped but none
    after -- while the product path (Model.train under torchrun) kept its graphs until interpreter exit and never destroyed the group at all.
    The order, made explicit:
      1. drop every CapturedStep (the graphs with the captured collectives and their private memory pools);
      2. close the GradBuckets (hooks off; finish() / exchange() leave no Work handle behind);
      3. gc.collect(): a graph held by a reference cycle is otherwise alive until some later collection;
      4. torch.cuda.synchronize(): every replay and collective has finished, the runtime has nothing in flight to defer a graph's release for;
      5. only then destroy_process_group().
    `captured`: a dict / list of CapturedStep (cleared in place); `buckets`: a GradBuckets or None.  Safe to call without a process group."""
    import gc
    import torch.distributed as dist
    if captured is not None:
        captured.clear()
    if buckets is not None:
        buckets.close()
    gc.collect()
    if torch.cuda.is_available():
        torch.cuda.synchronize()
    if destroy and dist.is_available() and dist.is_initialized():
        dist.destroy_process_group()
        global _OWNS_PROCESS_GROUP
        _OWNS_PROCESS_GROUP = False


def make_dataloader(dataset, args):
    """DataLoader of the reference (utils/parser.py:98-104); under torchrun every rank reads its own shard of the dataset
    (DistributedSampler: same shuffle seed on all ranks, disjoint indices), batch_size stays the PER-GPU batch."""
    workers = int(args.get("num_workers", 0) or 0)
    rank, world, _ = dist_env()
    shuffle = bool(args.get("shuffle", False))
    sampler = None
    if world > 1:
        from torch.utils.data.distributed import DistributedSampler
        sampler = DistributedSampler(dataset, num_replicas=world, rank=rank, shuffle=shuffle, drop_last=False)
        shuffle = False
    return DataLoader(dataset, batch_size=args["batch_size"], shuffle=shuffle, sampler=sampler, num_workers=workers,
                      pin_memory=torch.cuda.is_available())


# ---- logging ----------------------------------------------------------------------------------------------------------
class RunLogger:
    """runs/<name>/<timestamp>/{test,train}.{csv,jsonl} + summary.json (utils/logger.py:42-166), test rows only."""

    def __init__(self, config):
        log = config.get("logging") or {}
        self.enabled = bool(log.get("enabled", False))
        self._dir = None
        self._csv = {}
        if self.enabled:
            self._dir = os.path.join(log.get("root_dir", "runs"), str(config.get("name") or "run"), time.strftime("%Y%m%d_%H%M%S"))
            os.makedirs(self._dir, exist_ok=True)
            if log.get("save_config_copy", True):
                with open(os.path.join(self._dir, "config.json"), "w") as f:
                    json.dump(config, f, indent=2)

    def run_dir(self):
        return self._dir

    def log(self, split, row):
        if not self.enabled:
            return
        with open(os.path.join(self._dir, f"{split}.jsonl"), "a") as f:
            f.write(json.dumps(row) + "\n")
        path = os.path.join(self._dir, f"{split}.csv")
        new = split not in self._csv
        if new:
            self._csv[split] = list(row.keys())
        with open(path, "a", newline="") as f:
            w = csv.DictWriter(f, fieldnames=self._csv[split], extrasaction="ignore")
            if new:
                w.writeheader()
            w.writerow(row)

    def summary(self, data):
        if self.enabled:
            with open(os.path.join(self._dir, "summary.json"), "w") as f:
                json.dump(data, f, indent=2)

    def generate_plots(self):
        """loss curves of train.csv next to it (utils/logger.py:168-185 does the same, also best effort)"""
        if not self.enabled or "train" not in self._csv:
            return
        try:
            import matplotlib
            matplotlib.use("Agg")
            import matplotlib.pyplot as plt
            with open(os.path.join(self._dir, "train.csv")) as f:
                rows = [r for r in csv.DictReader(f) if r.get("type") == "epoch"]
            keys = [k for k in (rows[0] if rows else {}) if k.startswith("loss_")]
            if not keys:
                return
            fig, ax = plt.subplots(figsize=(7, 4))
            for k in keys:
                ax.plot([int(r["epoch"]) for r in rows], [float(r[k]) for r in rows], label=k[5:])
            ax.set_xlabel("epoch")
            ax.set_ylabel("training loss")
            ax.legend()
            fig.savefig(os.path.join(self._dir, "loss_curves.png"), dpi=120, bbox_inches="tight")
            plt.close(fig)
        except Exception as e:  # noqa: BLE001  (plots never fail a run)
            warnings.warn(f"loss curves not written: {e}")

    def close(self):
        self._csv.clear()


# ---- losses (utils/loss_factory.py:106-235; terms that need downloaded networks are skipped) -----------------------------------
class LossPipeline:
    """The reference's LossPipeline (utils/loss_factory.py:24-55) over the network-free terms, evaluated and
    differentiated by ONE HIP call (csrc/loss.hip): `pipeline(outputs, targets) -> (total, values)` where total is
    differentiable w.r.t. outputs and values[k] is term k unweighted (device tensor, values[-1] = total)."""

    def __init__(self, terms):
        self.terms = terms                       # [(name, weight, param)]
        self.names = [n for n, _, _ in terms]

    def __len__(self):
        return len(self.terms)

    def __call__(self, outputs, targets):
        from . import pipeline as PL
        return PL.fused_loss(outputs, targets, self.terms)


def build_losses(loss_cfg):
    if not loss_cfg or not loss_cfg.get("enabled", True):
        loss_cfg = {"terms": [{"name": "mse", "weight": 1.0}]}      # utils/loss_factory.py:120-122
    terms = []
    for t in loss_cfg.get("terms") or []:
        name, weight, args = t["name"], float(t.get("weight", 1.0)), (t.get("args") or {})
        if name in ("mse", "l1", "ssim"):
            param = 0.0
        elif name == "charbonnier":
            param = float(args.get("eps", 1e-3))
        elif name == "gradient_l1":
            param = 1.0 if bool(args.get("to_gray", False)) else 0.0
        elif name in ("vgg_perceptual", "lpips"):
            warnings.warn(f"loss term '{name}' needs downloaded network weights and is skipped on the offline MI355X path")
            continue
        else:
            raise ValueError(f"Unknown loss term: {name}")
        terms.append((name, weight, param))
    return LossPipeline(terms)


# ---- model harness ------------------------------------------------------------------------------------------------------
class Model:
    """Same constructor and entry points as models.model.Model (models/model.py:26, models/base.py:12,35-42)."""

    def __init__(self, network, *, config, dataloader, logger=None):
        self.config, self.phase = config, config["phase"]
        sect = config[self.phase]
        self.rank, self.world, local = init_distributed()
        self.device = torch.device(sect["device"])
        if self.world > 1 and self.device.type == "cuda" and self.device.index is None:
            self.device = torch.device("cuda", local)          # one process per GPU
        cap_cpu_threads()
        if self.device.type == "cuda" and torch.cuda.is_available():
            bind_to_gpu_numa(self.device.index if self.device.index is not None else torch.cuda.current_device())   # pinned batches stay on the GPU's socket
        self.model_path, self.model_name = sect["model_path"], sect["model_name"]
        test = config.get("test") or {}
        self.is_dataset_paired = bool((test.get("dataset") or {}).get("is_paired", True))
        self.dataloader, self.logger = dataloader, logger
        self.network = network.to(self.device)
        self.postproc_cfg = config.get("post_processing") or {"enabled": False}
        self.save_cfg = dict(config.get("save_outputs") or {})
        self.save_cfg.setdefault("output_dir", test.get("output_images_path", "outputs/"))
        self.save_cfg.setdefault("save_raw", False)
        self.save_cfg.setdefault("save_postprocessed", True)
        self.save_cfg.setdefault("raw_prefix", "raw_")
        self.save_cfg.setdefault("post_prefix", self.save_cfg.get("prefix", "output_"))
        ev = config.get("evaluation") or {}
        self.eval_on_raw = bool(ev.get("raw", True))
        self.eval_on_post = bool(ev.get("postprocessed", bool(self.postproc_cfg.get("enabled", False))))
        mcfg = config.get("metrics") or {"enabled": False}
        wanted = [m["name"] for m in (mcfg.get("items") or [])] if mcfg.get("enabled", False) else []
        self.metric_names = [m for m in wanted if m in ("psnr", "ssim")]
        for m in wanted:
            if m not in ("psnr", "ssim"):
                warnings.warn(f"metric '{m}' needs downloaded network weights and is skipped on the offline MI355X path")
        self.results = {}

    # -- reference entry points --
    def train(self):
        since = time.time()
        self.train_step()
        t = time.time() - since
        print(f"Training completed in {t // 60:.0f}m {t % 60:.0f}s")

    def train_step(self):
        """models/model.py:138-227: Adam(lr), loss pipeline, best-on-train-loss checkpoint.  Under fp16 storage
        (MDIE_PRECISION=fp16, the reference's own autocast dtype) the loss goes through torch's GradScaler exactly as at
        models/model.py:31,164-166 -- activation GRADIENTS are stored in fp16 and would underflow unscaled; fp32 / bf16 need none."""
        import torch.distributed as dist
        from . import train as T
        tr = self.config["train"]
        n_epoch, lr = int(tr["n_epoch"]), float(tr["lr"])
        losses = build_losses(self.config.get("loss"))
        if len(losses) == 0:
            raise ValueError("training needs at least one usable loss term")
        scaler = torch.amp.GradScaler("cuda", enabled=str(getattr(self.network, "precision", "fp32")).lower() in ("fp16", "f16", "float16", "half"))
        distributed = dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1
        # One hipGraph per batch shape (train.CapturedStep; MDIE_TRAIN_GRAPH=0 turns it off): forward + loss + backward, and the
        # Adam step too when nothing has to happen between backward and step (no gradient exchange, no GradScaler check).
        # (auto: only while a step is launch-bound -- measured on MI355X, bf16, B = 8, round 3: 256x256 6.2 -> 3.8 ms per step as a
        #  graph; 512x512 7.9 eager -- weight gradients overlap the backward chain on a side stream there, train._wgrad -- 8.4 as a graph)
        mode = os.environ.get("MDIE_TRAIN_GRAPH", "auto") if self.device.type == "cuda" else "0"
        # (round 4: the side stream is opt-in, MDIE_TRAIN_WGRAD_STREAM=1 -- without it the graph is at least as fast at every size, 8.28 against
        #  8.30 ms at 8x512x512, and takes the ~5.8 ms of host work per eager step off the critical path altogether: auto = always)
        want_graph = (lambda x: mode == "1" or (mode == "auto" and (not T.WGRAD_STREAM or x.shape[0] * x.shape[2] * x.shape[3] <= 8 * 384 * 384)))
        whole = mode != "0" and not scaler.is_enabled()      # the Adam step rides in the graph too (with the gradient exchange in front of it when distributed)
        # (fused=True: torch's single-kernel Adam -- the same update as the reference's default foreach form, models/model.py:146,
        #  in 1 launch instead of 19: 0.28 -> 0.06 ms of a 10 ms step; CPU runs keep the default)
        opt = torch.optim.Adam(self.network.parameters(), lr=lr, capturable=whole, **({"fused": True} if self.device.type == "cuda" else {}))
        captured = {}
        if distributed:
            # identical replicas: rank 0's parameters and buffers (the seed already makes them equal; this makes it certain)
            for t in list(self.network.parameters()) + list(self.network.buffers()):
                dist.broadcast(t.data, src=0)
        # Data parallel: the bucketed all-reduce ALWAYS overlaps backward (models/model.py:164-166 is the step being sharded).  In a captured
        # step the collectives are branches of the graph (train.CapturedStep, `buckets=`); in an eager step they are issued by the buckets'
        # hooks during backward.  There is no "exchange after a finished backward" on this path (round 4 ran five collectives behind the
        # replay).  Which form a distributed step takes: MDIE_DDP_CAPTURE = 0 (default) | 1 | auto.
        #   0     eager steps, the collectives issued by the buckets' hooks during backward: the DEFAULT under world > 1, because it is the only
        #         form that has ever run on more than one rank (two-rank gloo tests; the 1-rank RCCL test cannot see a wrong order between the
        #         gradient kernels, the communication stream's fork / join and the captured Adam step: an in-place all-reduce over one rank is
        #         the identity).
        #   1     every step captured, the collectives as branches of the graph (train.CapturedStep, `buckets=`).
        #   auto  captured below 8 x 384 x 384 input pixels, eager above.  That threshold was derived on ONE rank (profiles/r05c_ddp_forms.txt,
        #         bf16, B = 8: 512x512 eager + hooks 8.69 ms against 8.89 in the graph -- each of the five fork-joins costs a replay ~80 us;
        #         256x256 6.97 host-bound against 4.22), i.e. from host and launch overheads only: whether the overlap hides any exchange time
        #         is UNMEASURED until a multi-GPU box runs tools/bench_train.py --gpus N.
        n_buckets = int(os.environ.get("MDIE_DDP_BUCKETS", "4"))
        buckets = T.GradBuckets(self.network.parameters(), n_buckets=n_buckets) if distributed else None
        ddp_mode = os.environ.get("MDIE_DDP_CAPTURE", "0")
        if buckets is not None:
            small = (lambda x: x.shape[0] * x.shape[2] * x.shape[3] < 8 * 384 * 384)
            graph_ok = want_graph
            want_graph = (lambda x: graph_ok(x) and (ddp_mode == "1" or (ddp_mode == "auto" and small(x))))
        self.exchange_mode = None if buckets is None else {"0": "hooks (eager steps, overlapped with backward)", "1": "in-graph (captured steps)",
                                                           "auto": "in-graph below 8x384x384 input pixels, hooks above"}.get(ddp_mode, ddp_mode)
        self.n_buckets = None if buckets is None else len(buckets.buckets)
        best = float("inf")
        self.history = []
        try:
            for epoch in range(n_epoch):
                best = self._train_epoch(epoch, n_epoch, lr, losses, scaler, opt, captured, buckets, want_graph, whole, distributed, best)
        finally:
            # graphs first (under world > 1 they hold captured collectives of the communicator), then the buckets (hooks off, the training
            # Functions stop writing into this instance's slices), collected and synchronised -- also when a step raised; the process group
            # itself is host.run's to destroy (shutdown_distributed, the same order)
            shutdown_distributed(captured, buckets, destroy=False)
        return self.history

    def _train_epoch(self, epoch, n_epoch, lr, losses, scaler, opt, captured, buckets, want_graph, whole, distributed, best):
        import torch.distributed as dist
        from . import train as T
        t0 = time.time()
        self.network.train()
        if hasattr(getattr(self.dataloader, "sampler", None), "set_epoch"):
            self.dataloader.sampler.set_epoch(epoch)       # a different shard shuffle every epoch
        sums, n = {}, 0
        for inputs, targets in self.dataloader:
            x, y = self._to_device(inputs), self._to_device(targets)
            if want_graph(x):
                key = (tuple(x.shape), tuple(y.shape))
                if key not in captured:
                    captured[key] = T.CapturedStep(self.network, losses, opt if whole else None, x, y,
                                                   scale_fn=scaler.scale if scaler.is_enabled() else None, buckets=buckets)
                values = captured[key](x, y)
                if not whole:
                    scaler.step(opt)
                    scaler.update()
            else:
                opt.zero_grad(set_to_none=True)
                out = self.network(x)
                total, values = losses(out, y)
                scaler.scale(total).backward()
                if buckets is not None:      # averaged gradients: RCCL all-reduce launched from the grad hooks during backward
                    buckets.finish()
                scaler.step(opt)         # (unscales, skips the step on inf/nan; plain opt.step() when disabled)
                scaler.update()
            vals = values.cpu().tolist()  # one sync per step
            for k, v in zip(losses.names + ["total"], vals):
                sums[k] = sums.get(k, 0.0) + v
            n += 1
        if distributed:   # epoch means over ALL ranks' batches, so every rank takes the same checkpoint decision
            keys = sorted(sums)
            tot = torch.tensor([sums[k] for k in keys] + [float(n)], dtype=torch.float64, device=self.device)
            dist.all_reduce(tot)
            sums, n = {k: float(v) for k, v in zip(keys, tot[:-1].tolist())}, int(tot[-1].item())
        avg = {k: v / max(1, n) for k, v in sums.items()}
        main_rank = not distributed or dist.get_rank() == 0
        if avg["total"] < best:
            best = avg["total"]
            if main_rank:
                self.save_model(self.network)
        self.history.append(avg)
        if main_rank:
            print(f"Epoch [{epoch + 1}/{n_epoch}] Train total: {avg['total']:.4f} | " +
                  ", ".join(f"{k}: {v:.4f}" for k, v in avg.items() if k != "total") + f" | best: {best:.4f}")
        if self.logger is not None and main_rank:
            row = {"type": "epoch", "epoch": epoch + 1, "epoch_time_sec": time.time() - t0, "lr": lr, "best_loss_so_far": best}
            row.update({f"loss_{k}": v for k, v in avg.items()})
            self.logger.log("train", row)
        return best

    def test(self):
        self.test_step()

    def save_model(self, model):
        os.makedirs(self.model_path, exist_ok=True)
        torch.save(model.state_dict(), os.path.join(self.model_path, self.model_name))

    # -- helpers --
    def _to_device(self, batch):
        from . import pipeline as PL
        b = batch.to(self.device, non_blocking=True)
        return PL.feed_uint8(b) if b.dtype == torch.uint8 else b.float()

    def _save(self, u8_hwc, start, prefix):
        out_dir = self.save_cfg["output_dir"]
        os.makedirs(out_dir, exist_ok=True)
        fmt = str(self.save_cfg.get("format", "png")).lower()
        arr = u8_hwc.cpu().numpy()
        hw = self.save_cfg.get("resize_hw")                       # [h, w] or None (models/model.py:77,86-87)
        for i in range(arr.shape[0]):
            img = Image.fromarray(arr[i])
            if hw is not None:
                img = img.resize((int(hw[1]), int(hw[0])), Image.BILINEAR)
            img.save(os.path.join(out_dir, f"{prefix}{start + i + 1}.{fmt}"))   # 1-based, no padding: the reference's names (:90)

    def _metrics(self, out, tgt):
        from . import pipeline as PL
        if not self.metric_names:
            return {}
        v = PL.psnr_ssim(out, tgt).cpu()  # one D2H sync per evaluated tensor, not one per term
        vals = {"psnr": float(v[0]), "ssim": float(v[1])}
        return {k: vals[k] for k in self.metric_names}

    def test_step(self):
        from . import pipeline as PL
        path = os.path.join(self.model_path, self.model_name)
        if os.path.exists(path):
            self.network.load_state_dict(torch.load(path, map_location="cpu"))
        elif os.environ.get("MDIE_ALLOW_UNTRAINED") == "1":          # explicit opt-in only (benchmarks on synthetic weights)
            warnings.warn(f"checkpoint {path} not found: evaluating the network's current weights (MDIE_ALLOW_UNTRAINED=1)")
        else:
            raise FileNotFoundError(f"checkpoint {path} not found (the reference's test phase loads it unconditionally, models/model.py:230-231)")
        self.network.eval()
        sums = {"raw": {}, "post": {}}
        losses = build_losses(self.config.get("loss")) if self.is_dataset_paired else None     # models/model.py:257,264
        n_batches = n_images = 0
        max_save = self.save_cfg.get("max_images")
        t0 = time.time()
        with torch.no_grad():
            for batch in self.dataloader:
                inputs, targets = (batch if self.is_dataset_paired else (batch, None))
                x = self._to_device(inputs)
                t = self._to_device(targets) if targets is not None else None
                raw = self.network(x)
                want_post = bool(self.postproc_cfg.get("enabled", False))
                saving = self.save_cfg.get("enabled", True) and (max_save is None or n_images < max_save)
                pp, pp_u8 = PL.apply_postprocessing(raw, self.postproc_cfg, want_uint8=True) if want_post else (raw, None)
                if t is not None:
                    for stage, on, img in (("raw", self.eval_on_raw, raw), ("post", self.eval_on_post and want_post, pp)):
                        if not on:
                            continue
                        vals = dict(self._metrics(img, t))
                        if losses is not None and len(losses):      # the loss pipeline is evaluated in the test phase too
                            lv = losses(img, t)[1].cpu().tolist()
                            vals.update({f"loss_{k}": v for k, v in zip(losses.names + ["total"], lv)})
                        for k, v in vals.items():
                            sums[stage][k] = sums[stage].get(k, 0.0) + v
                if saving:
                    if self.save_cfg["save_raw"]:
                        self._save(PL.to_uint8_hwc(raw), n_images, self.save_cfg["raw_prefix"])
                    if self.save_cfg["save_postprocessed"]:
                        self._save(pp_u8 if pp_u8 is not None else PL.to_uint8_hwc(pp), n_images, self.save_cfg["post_prefix"])
                n_images += raw.shape[0]
                n_batches += 1
                if max_save is not None and n_images >= max_save:
                    break
        denom = max(1, n_batches)  # per-BATCH means, as the reference reports them (models/model.py:289-336)
        self.results = {stage: {k: v / denom for k, v in d.items()} for stage, d in sums.items()}
        self.results["n_images"], self.results["seconds"] = n_images, time.time() - t0
        for stage in ("raw", "post"):
            if self.results[stage]:
                print(f"[TEST {stage.upper()}] " + ", ".join(f"{k}: {v:.4f}" for k, v in self.results[stage].items()))
        if self.logger is not None:
            row = {"split": "test", "n_images": n_images}
            row.update({f"{s}_{k}": v for s in ("raw", "post") for k, v in self.results[s].items()})
            self.logger.log("test", row)
            self.logger.summary(self.results)
        return self.results


# ---- the driver ---------------------------------------------------------------------------------------------------------
def seed_everything(seed=42):
    """python / numpy / torch generators (the reference seeds 42 before every run, utils/reproducibility.py:6-20; the HIP
    kernels themselves are deterministic by construction: ordered reductions, counter-based dropout)"""
    import random
    random.seed(seed)
    np.random.seed(seed)
    torch.manual_seed(seed)
    if torch.cuda.is_available():
        torch.cuda.manual_seed_all(seed)


def run(config):
    """One run of the phase a parsed config names (`load_config(path, phase)`): the objects the config describes are built
    by the ["module", "Class"] factory in dependency order -- data, loader, network, harness -- and the harness method of
    the same name as the phase is called.  Returns the harness (its `.results` / `.history` hold what was measured)."""
    phase = config["phase"]
    if phase not in ("train", "test"):
        raise ValueError(f"phase must be 'train' or 'test', got {phase!r}")
    seed_everything(int(config.get("seed") or 42))
    section, model_cfg = config[phase], config["model"]
    log = RunLogger(config)
    if log.run_dir():
        print(f"run directory: {log.run_dir()}")
    loader = make_dataloader(instantiate(section["dataset"], default_module="data", kind="Dataset"), section["dataloader"]["args"])
    network = instantiate(model_cfg["networks"][0], default_module="models.network", kind="Network")
    harness_spec = {"name": model_cfg["which_model"]["name"], "args": dict(model_cfg["which_model"].get("args") or {})}
    harness = instantiate(harness_spec, network, default_module="models.model", kind="Model", config=config, dataloader=loader, logger=log)
    try:
        getattr(harness, phase)()
        if phase == "train":
            log.generate_plots()
    finally:
        log.close()
        if _OWNS_PROCESS_GROUP:      # this process joined the group in Model.__init__ (torchrun / self_launch): take it down, in order
            shutdown_distributed()
    return harness


def cli(argv=None):
    """`run.py -c config/<task>.json -p train|test`: the command line of the reference's run.py:37-53"""
    import argparse
    ap = argparse.ArgumentParser(description="CDAN restoration on the MI355X engine, driven by the reference's config files")
    ap.add_argument("-c", "--config", required=True, help="JSON config (// comments allowed), e.g. config/low_light.json")
    ap.add_argument("-p", "--phase", choices=("train", "test"), default="train")
    a = ap.parse_args(argv)
    return run(load_config(a.config, a.phase))
