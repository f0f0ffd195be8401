"""``captioning.modules.loss_wrapper.LossWrapper`` of the reference (captioning/modules/loss_wrapper.py:6-355), UIC branch."""
from boficap_amd.loss_wrapper import LossWrapper  # noqa: F401
