"""Option files of a run, as the reference writes them (options/base_options.py:116-149): ``<ckpt_dir>/<name>/opt.pkl`` is
the pickled option namespace, ``opt.txt`` a readable listing; ``--continue_training`` reads the pickle back and lets the
stored values override the defaults of every option except ``name`` / ``load_model_name``.

The pickle holds a plain namespace of python values (paths, numbers, lists, a torch.device).  ``load_options`` unpickles
files THIS package (or the user's own reference run) wrote -- it is never pointed at files shipped inside a repository."""
import pickle
from pathlib import Path


def option_file_path(opt, make_dir=False):
    expr_dir = Path(opt.ckpt_dir) / opt.name
    if make_dir:
        expr_dir.mkdir(parents=True, exist_ok=True)
    return expr_dir / "opt.pkl"


def save_options(opt, defaults=None):
    """opt.txt (one ``key: value`` line per option, ``[default: ...]`` noted where ``defaults`` -- a mapping -- differs) and
    opt.pkl (base_options.py:122-133)."""
    path = option_file_path(opt, make_dir=True)
    with path.with_suffix(".txt").open("w") as f:
        for k, v in sorted(vars(opt).items()):
            comment = ""
            if defaults is not None and k in defaults and v != defaults[k]:
                comment = "\t[default: %s]" % str(defaults[k])
            f.write("{:>25}: {:<30}{}\n".format(str(k), str(v), comment))
    with path.with_suffix(".pkl").open("wb") as f:
        pickle.dump(opt, f)
    return path


def load_options(opt):
    """The stored namespace of a run: the run's own ``opt.pkl`` when continuing, else ``opt.load_from_opt_file``
    (base_options.py:143-149)."""
    path = option_file_path(opt) if getattr(opt, "continue_training", False) else Path(opt.load_from_opt_file)
    with path.open("rb") as f:
        return _OptionUnpickler(f).load()


class _OptionUnpickler(pickle.Unpickler):
    """An option namespace is plain data: only the handful of constructors such a file needs are resolvable -- anything else in the
    stream (an arbitrary callable) is refused instead of imported and called."""
    _ALLOWED = {("argparse", "Namespace"), ("types", "SimpleNamespace"), ("pathlib", "PosixPath"), ("pathlib", "PurePosixPath"),
                ("pathlib", "Path"), ("torch", "device"), ("collections", "OrderedDict"), ("builtins", "set"),
                ("builtins", "frozenset"), ("builtins", "complex"), ("builtins", "slice"), ("builtins", "range")}

    def find_class(self, module, name):
        if (module, name) in self._ALLOWED:
            return super().find_class(module, name)
        raise pickle.UnpicklingError(f"option file refers to {module}.{name}: not an option value")


def update_options_from_file(opt, explicit=(), keep=("name", "load_model_name", "continue_training", "ckpt_dir", "device")):
    """Apply the stored run's values onto ``opt``, as the reference's two-pass parse does by installing them as parser DEFAULTS
    (base_options.py:135-141): ``name`` / ``load_model_name`` are never taken from the file, and anything given on the command
    line of the continuing run still wins -- ``explicit`` names those options (the reference's argparse knows them by itself; a
    caller that builds ``opt`` directly passes e.g. ``explicit=("num_epochs", "lr")``), ``keep`` the ones this package always
    treats as given (``continue_training``, where the run lives, the device).  Returns ``opt``."""
    old = load_options(opt)
    given = set(keep) | set(explicit)
    for k, v in vars(old).items():
        if k not in given and hasattr(opt, k):
            setattr(opt, k, v)
    return opt
