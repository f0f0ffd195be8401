"""``captioning.models.setup(opt)`` -- the reference's constructor entry point
(/root/reference/captioning/models/__init__.py:14-24), backed by boficap_amd."""
from boficap_amd.transformer_model import TransformerModel, setup  # noqa: F401
