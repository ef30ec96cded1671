"""CPU oracle of the MixQ quantized-Linear path — TEST INFRASTRUCTURE ONLY (see oracle/mixq_oracle.c)."""
