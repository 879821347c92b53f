"""Drop-in for the reference's ``model`` package (model/CPPN.py, model/Temporal.py)."""
