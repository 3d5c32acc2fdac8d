"""Residue templates for the internal-coordinate (z-matrix) reconstruction.

Same content as the reference's `core_atoms` / `atom_order_list`
(reference utils/utils_ic.py:6-83), written as one compact spec and expanded at
import.  Slot layout per residue is fixed: [O, N, C, CA, side-chain atom 0..9];
`atom_order_list[res][i]` names, for side-chain atom i, the three already placed
slots (bonded, angle, dihedral partner) it is built from.
"""

# name : side-chain atoms in placement order | one hex triplet per side-chain atom
_SPEC = """
ALA CB | 123
ARG CB CG CD NE CZ NH1 NH2 | 123 234 345 456 567 678 789
ASP CB CG OD1 OD2 | 123 234 345 456
ASN CB CG OD1 ND2 | 123 234 345 456
CYS CB SG | 123 234
GLU CB CG CD OE1 OE2 | 123 234 345 456 567
GLN CB CG CD OE1 NE2 | 123 234 345 456 567
GLY |
HIS CB CG CD2 ND1 NE2 CE1 | 123 234 345 345 756 568
ILE CB CG2 CG1 CD1 | 123 234 345 346
LEU CB CG CD1 CD2 | 123 234 345 456
LYS CB CG CD CE NZ | 123 234 345 456 567
MET CB CG SD CE | 123 234 345 456
PHE CB CG CD1 CE1 CZ CD2 CE2 | 123 234 345 456 567 345 459
PRO CB CG CD | 123 134 431
SER CB OG | 123 234
THR CB OG1 CG2 | 123 234 345
TRP CB CG CD1 CD2 NE1 CE2 CZ2 CH2 CE3 CZ3 | 123 234 345 345 756 657 579 79a a97 97c
TYR CB CG CD1 CD2 CE2 CZ CE1 OH | 123 234 345 345 657 578 789 789
VAL CB CG1 CG2 | 123 234 345
TPO CB OG1 CG2 P OE1 OE2 OE3 | 123 234 234 645 457 457 457
SEP CB OG P OE1 OE2 OE3 | 123 234 345 456 456 456
"""

BACKBONE_SLOTS = ['O', 'N', 'C', 'CA']
MAX_SIDECHAIN = 10
SLOTS_PER_RESIDUE = 14

core_atoms = {}
atom_order_list = {}
for _line in _SPEC.strip().splitlines():
    _head, _trips = _line.split('|')
    _name, *_side = _head.split()
    core_atoms[_name] = BACKBONE_SLOTS + _side
    atom_order_list[_name] = [[int(c, 16) for c in t] for t in _trips.split()]
    assert len(atom_order_list[_name]) == len(_side) <= MAX_SIDECHAIN
