// ORACLE — test infrastructure, not product code.  Chapter-1 CPU Cornell tracer placeholder (filled in below).
