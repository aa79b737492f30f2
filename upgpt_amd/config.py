"""Config-driven construction — the reference's plugin mechanism (ldm/util.py:78-93):
every YAML node {target: dotted.path, params: {...}} is imported and constructed."""
import importlib


def to_plain(cfg):
    """OmegaConf-like containers -> plain dict/list (omegaconf is optional)."""
    try:
        from omegaconf import OmegaConf  # type: ignore
        if OmegaConf.is_config(cfg):
            return OmegaConf.to_container(cfg, resolve=True)
    except Exception:
        pass
    if isinstance(cfg, dict):
        return {k: to_plain(v) for k, v in cfg.items()}
    if isinstance(cfg, (list, tuple)):
        return [to_plain(v) for v in cfg]
    return cfg


def get_obj_from_str(string, reload=False):
    module, cls = string.rsplit(".", 1)
    mod = importlib.import_module(module, package=None)
    if reload:
        mod = importlib.reload(mod)
    return getattr(mod, cls)


def instantiate_from_config(config):
    if "target" not in config:
        if config in ("__is_first_stage__", "__is_unconditional__"):
            return None
        raise KeyError("Expected key `target` to instantiate.")
    params = config.get("params", dict())
    return get_obj_from_str(config["target"])(**(params if params is not None else dict()))


def load_config(path):
    """YAML -> plain dict (configs/deepfashion/bbox.yaml parses unchanged)."""
    import yaml
    with open(path) as f:
        return yaml.safe_load(f)


def default(val, d):
    if val is not None:
        return val
    return d() if callable(d) else d


def exists(x):
    return x is not None


def count_params(model, verbose=False):
    total = sum(p.numel() for p in model.parameters())
    if verbose:
        print(f"{model.__class__.__name__} has {total * 1.e-6:.2f} M params.")
    return total
