from .utils import replace_module_by_qmodule_deit  # noqa: F401
