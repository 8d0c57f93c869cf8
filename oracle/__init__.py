"""CPU oracle (test infrastructure only): see oracle/orc.h."""
