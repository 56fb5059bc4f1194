"""Alias of the reference's `model` package: re-exports of geoformer_amd (see ../README.md)."""
