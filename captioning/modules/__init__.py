"""Drop-in import path of the reference's ``captioning.modules`` for the bound+fill (UIC) path."""
