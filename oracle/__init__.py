"""TEST INFRASTRUCTURE ONLY: CPU oracle for the XQ-GAN quantizer hot path (see oracle/xq_oracle.c)."""
