// empty on purpose: everything the host build needs comes from oracle/ref_shim/prelude.h
