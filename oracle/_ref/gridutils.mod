﻿!mod$ v1 sum:6c3d806c99880ca0
module gridutils
real(8),parameter::symtol=9.99999982451670044181213370393379591405391693115234375e-15_8
contains
function symgrid(x,nx)
integer(4),intent(in)::nx
real(8),intent(in)::x(1_8:int(nx,kind=8))
logical(4)::symgrid
end
end
