"""CPU oracle: test infrastructure only (see apmf_oracle.c).  Never imported by the product."""
