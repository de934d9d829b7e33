"""Drop-in `run(argv)` tools for the two pipelines on the hot path, registered the way the reference
registers its own (module list handed to simppl: /root/reference/ugvc/__main__.py:42-56,103-105)."""
from . import filter_variants_pipeline, train_models_pipeline  # noqa: F401

MODULES = [filter_variants_pipeline, train_models_pipeline]
