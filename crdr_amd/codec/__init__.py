"""Host side of the entropy coder: CDF tables and the rANS binding (serial on the host, as in the reference)."""
