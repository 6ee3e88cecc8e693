"""Headless counterparts of pybo's demos (no plotting: `ezplot` is out of scope)."""
