"""`utils.parser` names the reference's run.py imports (utils/parser.py:28-104), backed by mdie_amd.host."""
from mdie_amd import host as _h

NoneDict = _h.Cfg
init_obj = _h.instantiate


def parse(args):
    return _h.load_config(args.config, args.phase)


def define_network(network_config):
    return _h.instantiate(network_config, default_module="models.network", kind="Network")


def define_dataset(dataset_config):
    return _h.instantiate(dataset_config, default_module="data", kind="Dataset")


def define_dataloader(dataset, dataloader_config):
    return _h.make_dataloader(dataset, dataloader_config)


def create_model(**cfg_model):
    spec = cfg_model["config"]["model"]["which_model"]
    network = cfg_model.pop("network")
    return _h.instantiate({"name": spec["name"], "args": dict(spec.get("args") or {})}, network, default_module="models.model",
                          kind="Model", **cfg_model)
