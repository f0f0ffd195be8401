"""Import-path shim: lets code written against the reference (``import captioning.models``) pick up
the MI355X implementation in ``boficap_amd`` unchanged.  Only the bound+fill hot path is provided."""
