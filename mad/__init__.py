"""Drop-in alias of the reference package layout (`from mad import MaD`); see mad_amd/."""
