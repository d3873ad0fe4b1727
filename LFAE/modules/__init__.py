"""see DM/__init__.py"""
