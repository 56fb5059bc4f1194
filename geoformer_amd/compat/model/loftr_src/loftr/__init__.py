"""alias package (see ../../../README.md)"""
