"""Drop-in for the reference's ``train`` helpers (model_helpers, data_helpers, proj_helpers)."""
