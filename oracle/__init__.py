"""CPU oracle — TEST INFRASTRUCTURE ONLY (see dsr_oracle.cpp header)."""
