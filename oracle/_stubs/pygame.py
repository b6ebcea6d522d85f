"""Empty stand-in for pygame (test infrastructure only): the reference imports it
at module scope for drawing (`physics/world.py:4` etc.) but never touches it when
run headless (`screen=None`)."""
