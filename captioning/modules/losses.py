"""``captioning.modules.losses`` of the reference, the criteria of the UIC path (captioning/modules/losses.py:29-179, 315-369)."""
from boficap_amd.loss_wrapper import LanguageModelCriterion_UIC, StructureLosses  # noqa: F401
