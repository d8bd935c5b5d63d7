"""ORACLE -- test infrastructure only (CPU restatement of the reference's arithmetic).
Never imported by the product package; see oracle/keras_ops.py header."""
