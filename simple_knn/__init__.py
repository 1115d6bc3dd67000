"""Shim package for the reference's `simple_knn` extension (litegs/scene/__init__.py:3); see simple_knn/_C.py."""
