"""Alias package: see dropin/README.md."""
