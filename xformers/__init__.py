"""Import shim: the reference only imports the TYPE `xformers.ops.AttentionBias` (qserve/utils/input_metadata.py:12,201);
no xformers arithmetic is on the path."""
